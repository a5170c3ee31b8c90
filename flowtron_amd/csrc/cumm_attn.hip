// Cumulative ("location-sensitive") attention of a teacher-forced flow as ONE C-ABI call per sequence and direction
// (reference flowtron.py:697-723 `run_cumm_attn_sequence`, :129-152 `AttentionConditioningLayer`, :544-592 `Attention.forward`).
//
// Frame i of the reference loop:
//     x_i    = [cumm_i ; prev_i]                         cumm_i = sum_{j<i} attn_j, prev_i = attn_{i-1}         [B,2,L]
//     cond_i = sigmoid(conv_k3(relu(conv_k5(x_i))))       2 -> 32 -> E channels                                  [L,B,E]
//     K_i    = (text . cond_i) W_key^T                    the key projection is redone EVERY frame               [L,B,A]
//     e_il   = v . tanh(Q_i + K_i[l]) / temperature ; attn_i = softmax over l < in_len ; logprob_i = log(attn_i + 1e-8)
//     ctx_i  = attn_i V
// The frames are sequentially dependent through (cumm, prev), and the work of a frame is dominated by a [L B, E] x [E, A]
// GEMM (4.1 GFLOP at B 32, L 157: the whole decoder of the default model costs less per frame), so the loop itself is what
// has to stay cheap.  The drop-in module used to walk the frames in Python (ten autograd nodes, ten allocations and ~250 us of
// host time per frame: 1.78 s per training step at BASELINE configs[1]'s shape); here the library walks them: every buffer is
// carved once from a caller-provided workspace, a frame is 9 (forward) / 23 (backward) kernel launches enqueued back to back
// on the caller's stream -- the existing im2col / GEMM / activation kernels plus the fused score kernels below -- and
// backward re-derives cond_i from the saved cumm_i instead of keeping T x [L,B,E] tensors.
//
// Saved for backward: cumm_all [T,B,L] (cumulative attention BEFORE frame i), kproj_all [T][L*B][A] (the projected keys of every
// frame: 11 GB per flow at T 862 -- sized for 288 GB of HBM; recomputing them would repeat the dominant GEMM), attn itself.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr float C2 = 2.8853900817779268f;      // 2 log2(e): tanh(x) = 1 - 2 / (2^(C2 x) + 1), as attention.hip
__device__ __forceinline__ float rsig(float x) { return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x) + 1.0f); }

struct Buf {            // carved from the workspace (floats)
    float *s2, *col1, *h1, *col2, *cond, *km, *esc;                             // forward chain of one frame (esc: scores / dp [B][L])
    float *dkp, *dkm, *dcond, *dcol2, *dh1, *dcol1, *ds2, *g_prev, *g_cumm;      // backward only
    int* full_lens;                                                             // [B] = L (Conv1d pads at the ends only)
    void* gemm_work; size_t gemm_work_bytes;
    size_t total;
};

inline size_t up64(size_t v) { return (v + 63) & ~size_t(63); }

Buf carve(void* base, int L, int B, int E, int A, int NF, int K1, int K2, bool bwd) {
    Buf b{};
    size_t off = 0;
    const size_t R = (size_t)L * B;
    auto take = [&](size_t n) { float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr; off += up64(n * 4); return p; };
    b.s2 = take(R * 2); b.col1 = take(R * 2 * K1); b.h1 = take(R * NF); b.col2 = take(R * NF * K2); b.cond = take(R * E); b.km = take(R * E); b.esc = take(R);
    if (bwd) {
        b.dkp = take(R * A); b.dkm = take(R * E); b.dcond = take(R * E); b.dcol2 = take(R * NF * K2); b.dh1 = take(R * NF);
        b.dcol1 = take(R * 2 * K1); b.ds2 = take(R * 2); b.g_prev = take((size_t)B * L); b.g_cumm = take((size_t)B * L);
    }
    b.full_lens = reinterpret_cast<int*>(take((size_t)B));
    // image GEMM scratch for the largest call ([R, E] x [E, A] and its transposes): two bf16 images, padded to 256
    const size_t img = ((R + 32 + 255) / 256 * 256) * ((size_t)((E > A ? E : A) + 255) / 256 * 256) * 2;
    b.gemm_work_bytes = 2 * img + 2 * (size_t)((E + 255) / 256 * 256) * ((A + 255) / 256 * 256) * 2 + 4096;
    b.gemm_work = take(b.gemm_work_bytes / 4 + 64);
    if (b.gemm_work) b.gemm_work = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(b.gemm_work) + 255) & ~uintptr_t(255));
    b.total = off + 256;
    return b;
}

__global__ void fill_int_k(int* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// s2[l][b][0] = cumm[b][l], s2[l][b][1] = prev[b][l] (prev == nullptr: zeros); cumm row stride = L, prev row stride = prev_ld
__global__ void stack2_k(const float* __restrict__ cumm, const float* __restrict__ prev, long prev_ld, float* __restrict__ s2, int L, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * B) return;
    const int l = i / B, b = i - l * B;
    s2[2 * (size_t)i] = cumm[(size_t)b * L + l];
    s2[2 * (size_t)i + 1] = prev ? prev[(size_t)b * prev_ld + l] : 0.f;
}
// g_prev[b][l] = ds2[l][b][1];  g_cumm[b][l] += ds2[l][b][0]
__global__ void unstack_acc_k(const float* __restrict__ ds2, float* __restrict__ g_prev, float* __restrict__ g_cumm, int L, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * B) return;
    const int l = i / B, b = i - l * B;
    g_cumm[(size_t)b * L + l] += ds2[2 * (size_t)i];
    g_prev[(size_t)b * L + l] = ds2[2 * (size_t)i + 1];
}
__global__ void fma_acc_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = fmaf(a[i], b[i], out[i]);
}
// out[n] += sum_r x[r][n]: grid (ceil(N / 64), ceil(rows / 256)), a block sums a 256-row slab of 64 columns and adds it with one
// fp32 atomic per column (the output accumulates over the frames anyway)
__global__ __launch_bounds__(256) void colsum_acc_k(const float* __restrict__ x, long rows, int N, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * 256, r1 = r0 + 256 < rows ? r0 + 256 : rows;
    float s = 0.f;
    if (c < N) {
#pragma unroll 8
        for (long r = r0 + ry; r < r1; r += 4) s += x[r * N + c];
    }
    red[ry][threadIdx.x & 63] = s;
    __syncthreads();
    if (ry == 0 && c < N) atomicAdd(out + c, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// ---- scores / softmax / context / running sum of ONE frame, spread over the chip ------------------------------------------
// A frame has only B x L scores of A terms each (3.2 M tanh at B 32, L 157): small, but on the critical path 862 times per
// flow, so it is cut for PARALLELISM, not for launch count: scores in (b, 16-key) workgroups, softmax + context in
// (b, 64-channel) workgroups that each redo the 157-element softmax of their utterance (cheaper than another hand-off).

// e[b][l] = v . tanh(Q[i][b] + K_i[l][b]) / temperature, l < in_len: grid (ceil(L / 16), B), a wave takes 4 keys
__global__ __launch_bounds__(256) void cumm_scores_k(const float* __restrict__ Q, const float* __restrict__ kp, const float* __restrict__ v,
                                                     const int* __restrict__ in_lens, float* __restrict__ e,
                                                     int i, int B, int L, int A, float inv_temp) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = min(in_lens[b], L);
    const float* q = Q + ((size_t)i * B + b) * A;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int l = blockIdx.x * 16 + wave * 4 + j;
        if (l >= len) break;                                    // wave-uniform
        const float* k = kp + ((size_t)l * B + b) * A;
        float acc = 0.f, vsum = 0.f;
        for (int a = lane; a < A; a += 64) { const float va = v[a]; acc = fmaf(va, rsig(C2 * (q[a] + k[a])), acc); vsum += va; }
        acc = wave_sum(acc); vsum = wave_sum(vsum);
        if (lane == 0) e[(size_t)b * L + l] = (vsum - 2.f * acc) * inv_temp;
    }
}

// softmax of e[b][:] into LDS (every workgroup of utterance b does it for itself): ps[l] = p_l for l < len; returns nothing
__device__ __forceinline__ void softmax_row(const float* __restrict__ e, float* ps, float* red, int len, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int l = tid; l < len; l += 256) { const float x = e[l]; ps[l] = x; m = fmaxf(m, x); }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int l = tid; l < len; l += 256) { const float x = expf(ps[l] - m); ps[l] = x; s += x; }
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    s = (red[4] + red[5]) + (red[6] + red[7]);
    for (int l = tid; l < len; l += 256) ps[l] = ps[l] / s;
    __syncthreads();
}

// grid (ceil(A / 64), B): ctx[i][b][a0 .. a0+63] = sum_l p_l V[l][b][a]; workgroup x == 0 also writes attn, logprob, cumm_next
__global__ __launch_bounds__(256) void cumm_ctx_k(const float* __restrict__ e, const float* __restrict__ V, const int* __restrict__ in_lens,
                                                  const float* __restrict__ cumm, float* __restrict__ cumm_next,
                                                  float* __restrict__ attn, float* __restrict__ logprob, float* __restrict__ ctx,
                                                  int i, int T, int B, int L, int A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ps = sm;                    // [L]
    float* part = ps + ((L + 3) & ~3); // [4][64]
    __shared__ float red[8];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = min(in_lens[b], L);
    softmax_row(e + (size_t)b * L, ps, red, len, tid);
    if (blockIdx.x == 0) {
        const size_t row = ((size_t)b * T + i) * L;
        for (int l = tid; l < L; l += 256) {
            const float p = l < len ? ps[l] : 0.f;
            attn[row + l] = p;
            logprob[row + l] = logf(p + 1e-8f);
            if (cumm_next) cumm_next[(size_t)b * L + l] = cumm[(size_t)b * L + l] + p;
        }
    }
    const int a = blockIdx.x * 64 + lane;
    float c = 0.f;
    if (a < A) {
#pragma unroll 8
        for (int l = wave; l < len; l += 4) c = fmaf(ps[l], V[((size_t)l * B + b) * A + a], c);
    }
    part[wave * 64 + lane] = c;
    __syncthreads();
    if (wave == 0 && a < A) ctx[((size_t)i * B + b) * A + a] = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
}

// backward, part 1: dp[b][l] = external + carried + dctx[i][b] . V[l][b]   (grid (ceil(L / 16), B), a wave takes 4 keys)
__global__ __launch_bounds__(256) void cumm_dp_k(const float* __restrict__ V, const int* __restrict__ in_lens, const float* __restrict__ attn,
                                                 const float* __restrict__ dctx, const float* __restrict__ dattn,
                                                 const float* __restrict__ dlogprob, const float* __restrict__ g_prev,
                                                 const float* __restrict__ g_cumm, float* __restrict__ dp,
                                                 int i, int T, int B, int L, int A) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = min(in_lens[b], L);
    const float* dc = dctx + ((size_t)i * B + b) * A;
    const size_t row = ((size_t)b * T + i) * L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int l = blockIdx.x * 16 + wave * 4 + j;
        if (l >= len) break;
        const float* vl = V + ((size_t)l * B + b) * A;
        float acc = 0.f;
        for (int a = lane; a < A; a += 64) acc = fmaf(dc[a], vl[a], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            float d = acc + g_prev[(size_t)b * L + l] + g_cumm[(size_t)b * L + l];
            if (dattn) d += dattn[row + l];
            if (dlogprob) d += dlogprob[row + l] / (attn[row + l] + 1e-8f);
            dp[(size_t)b * L + l] = d;
        }
    }
}

// backward, part 2: grid (ceil(A / 64), B).  s_l = p_l (dp_l - sum_m p_m dp_m) / temperature (redone per workgroup), then for the
// workgroup's 64 channels: dK_i[l][b][a] = s_l v_a (1 - tanh^2), dQ[i][b][a] = sum_l of it, dv_a += sum_l s_l tanh,
// dV[l][b][a] += p_l dctx[i][b][a]; rows l >= in_len of dK_i are zeroed
__global__ __launch_bounds__(256) void cumm_score_bwd_k(const float* __restrict__ Q, const float* __restrict__ kp, const float* __restrict__ v,
                                                        const int* __restrict__ in_lens, const float* __restrict__ attn,
                                                        const float* __restrict__ dctx, const float* __restrict__ dp,
                                                        float* __restrict__ dQ, float* __restrict__ dkp, float* __restrict__ dV,
                                                        float* __restrict__ dv, int i, int T, int B, int L, int A, float inv_temp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ps = sm;                        // [L] attention
    float* ss = ps + ((L + 3) & ~3);       // [L] s_l
    float* part = ss + ((L + 3) & ~3);     // [2][4][64]
    __shared__ float red[4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = min(in_lens[b], L);
    const size_t row = ((size_t)b * T + i) * L;
    float s = 0.f;
    for (int l = tid; l < len; l += 256) { const float p = attn[row + l], d = dp[(size_t)b * L + l]; ps[l] = p; ss[l] = d; s = fmaf(p, d, s); }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    for (int l = tid; l < len; l += 256) ss[l] = ps[l] * (ss[l] - s) * inv_temp;
    __syncthreads();
    const int a = blockIdx.x * 64 + lane;
    float dq = 0.f, dva = 0.f;
    if (a < A) {
        const float va = v[a], q = Q[((size_t)i * B + b) * A + a], dc = dctx[((size_t)i * B + b) * A + a];
#pragma unroll 4
        for (int l = wave; l < len; l += 4) {
            const size_t o = ((size_t)l * B + b) * A + a;
            const float r = rsig(C2 * (q + kp[o]));
            const float g = ss[l] * va * (4.f * fmaf(-r, r, r));           // 1 - tanh^2 = 4 r (1 - r)
            dq += g;
            dkp[o] = g;
            dva = fmaf(ss[l], fmaf(-2.f, r, 1.f), dva);
            dV[o] = fmaf(ps[l], dc, dV[o]);
        }
        for (int l = len + wave; l < L; l += 4) dkp[((size_t)l * B + b) * A + a] = 0.f;
    }
    part[wave * 64 + lane] = dq;
    part[256 + wave * 64 + lane] = dva;
    __syncthreads();
    if (wave == 0 && a < A) {
        dQ[((size_t)i * B + b) * A + a] = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
        atomicAdd(dv + a, (part[256 + lane] + part[320 + lane]) + (part[384 + lane] + part[448 + lane]));
    }
}

int gemm(const float* A, const float* Bm, float* C, const float* bias, int M, int N, int K, long sAm, long sAk, long sBk, long sBn, long ldc,
         float beta, int act, int mode, int flags, const Buf& b, hipStream_t st) {
    ft_gemm_args a{};
    a.A = A; a.B = Bm; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.batch = 1;
    a.sAm = sAm; a.sAk = sAk; a.sBk = sBk; a.sBn = sBn; a.ldc = ldc; a.alpha = 1.f; a.beta = beta; a.act = act; a.mode = mode; a.flags = flags;
    const size_t need = ft_gemm_workspace_bytes(&a);
    if (need && need <= b.gemm_work_bytes) { a.work = b.gemm_work; a.work_bytes = need; }
    return ft_gemm(&a, st);
}

#define CK(x) do { int rc_ = (x); if (rc_ != FT_OK) return rc_; } while (0)

// the location convolutions of frame i: (cumm_i, prev_i) -> h1 (relu), col1, col2, cond (sigmoid), km = text . cond
int cond_chain(const ft_cumm_attn_args* a, const Buf& b, const float* cumm_i, const float* prev_i, hipStream_t st) {
    const int L = a->L, B = a->B, E = a->E, R = L * B, NF = a->NF;
    hipLaunchKernelGGL(stack2_k, dim3(cdiv(R, 256)), dim3(256), 0, st, cumm_i, prev_i, (long)a->T * L, b.s2, L, B);
    CK(ft_im2col(b.s2, b.col1, b.full_lens, L, B, 2, a->K1, st));
    CK(gemm(b.col1, a->w1, b.h1, a->b1, R, NF, 2 * a->K1, 2 * a->K1, 1, 1, 2 * a->K1, NF, 0.f, FT_ACT_RELU, a->mode, 0, b, st));
    CK(ft_im2col(b.h1, b.col2, b.full_lens, L, B, NF, a->K2, st));
    CK(gemm(b.col2, a->w2, b.cond, a->b2, R, E, NF * a->K2, NF * a->K2, 1, 1, NF * a->K2, E, 0.f, FT_ACT_SIGMOID, a->mode, 0, b, st));
    CK(ft_eltwise(a->text, b.cond, b.km, (int64_t)R * E, 1, st));
    return FT_OK;
}

int check(const ft_cumm_attn_args* a) {
    FT_CHECK_ARG(a && a->text && a->Q && a->V && a->w_key && a->v && a->w1 && a->b1 && a->w2 && a->b2 && a->in_lens);
    FT_CHECK_ARG(a->ctx && a->attn && a->logprob && a->cumm_all && a->kproj_all && a->work);
    FT_CHECK_ARG(a->T >= 1 && a->B >= 1 && a->L >= 1 && a->E >= 1 && a->A >= 1 && a->NF >= 1 && (a->K1 & 1) && (a->K2 & 1) && a->temperature > 0.f);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(a->work) % 256 == 0);
    FT_CHECK_ARG((size_t)(2 * a->L + 1024) * 4 <= 60 * 1024 && a->B <= 65535);
    return FT_OK;
}

// the fused one-launch-per-frame path of the 16-bit operand modes (cumm_fused.hip, compiled once per format).
// FT_CUMM_FUSED=0 keeps the launch chain below (the yardstick of tests/test_gpu_model.py); read per call.
bool fused_on() { const char* e = getenv("FT_CUMM_FUSED"); return !e || atoi(e) != 0; }
int fused_supported(const ft_cumm_attn_args* a) {
    if (!fused_on()) return 0;
    return a->mode == FT_F16 ? ftint_cummf_supported_f16(a) : ftint_cummf_supported(a);
}

}  // namespace

extern "C" size_t ft_cumm_attn_workspace_bytes(int T, int L, int B, int E, int A, int NF, int K1, int K2, int mode, int backward) {
    size_t n = carve(nullptr, L, B, E, A, NF, K1, K2, backward != 0).total;
    ft_cumm_attn_args a{};
    a.T = T; a.L = L; a.B = B; a.E = E; a.A = A; a.NF = NF; a.K1 = K1; a.K2 = K2; a.mode = mode;
    if (fused_supported(&a)) {
        const size_t f = mode == FT_F16 ? ftint_cummf_workspace_bytes_f16(T, L, B, E, A, backward) : ftint_cummf_workspace_bytes(T, L, B, E, A, backward);
        if (f > n) n = f;
    }
    return n;
}

extern "C" int ft_cumm_attn_fused(const ft_cumm_attn_args* a) { return a ? fused_supported(a) : 0; }
extern "C" int ft_cumm_attn_debug_prof(void* dev_buf) { ftint_cummf_debug_prof(dev_buf); ftint_cummf_debug_prof_f16(dev_buf); return FT_OK; }

extern "C" int ft_cumm_attn_fwd(const ft_cumm_attn_args* a, void* stream) {
    CK(check(a));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (fused_supported(a)) return a->mode == FT_F16 ? ftint_cummf_fwd_f16(a, st) : ftint_cummf_fwd(a, st);
    const int T = a->T, B = a->B, L = a->L, E = a->E, A = a->A, R = L * B;
    const Buf b = carve(a->work, L, B, E, A, a->NF, a->K1, a->K2, false);
    FT_CHECK_ARG(a->work_bytes >= b.total);
    hipLaunchKernelGGL(fill_int_k, dim3(cdiv(B, 256)), dim3(256), 0, st, b.full_lens, B, L);
    FT_CHECK_HIP(hipMemsetAsync(a->cumm_all, 0, sizeof(float) * (size_t)B * L, st));          // cumm_0 = 0
    const size_t lds = sizeof(float) * ((size_t)((L + 3) & ~3) + 256);
    for (int i = 0; i < T; ++i) {
        float* cumm_i = a->cumm_all + (size_t)i * B * L;
        const float* prev_i = i > 0 ? a->attn + (size_t)(i - 1) * L : nullptr;                  // attn[b][i-1][:], row stride T*L
        CK(cond_chain(a, b, cumm_i, prev_i, st));
        float* kp = a->kproj_all + (size_t)i * R * A;
        CK(gemm(b.km, a->w_key, kp, nullptr, R, A, E, E, 1, 1, E, A, 0.f, FT_ACT_NONE, a->mode, 0, b, st));
        hipLaunchKernelGGL(cumm_scores_k, dim3(cdiv(L, 16), B), dim3(256), 0, st, a->Q, kp, a->v, a->in_lens, b.esc, i, B, L, A, 1.0f / a->temperature);
        hipLaunchKernelGGL(cumm_ctx_k, dim3(cdiv(A, 64), B), dim3(256), lds, st, b.esc, a->V, a->in_lens, cumm_i,
                           i + 1 < T ? cumm_i + (size_t)B * L : nullptr, a->attn, a->logprob, a->ctx, i, T, B, L, A);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int ft_cumm_attn_bwd(const ft_cumm_attn_args* a, const float* dctx, const float* dattn, const float* dlogprob,
                                float* dQ, float* dV, float* dtext, float* dw_key, float* dv, float* dw1, float* db1, float* dw2, float* db2,
                                void* stream) {
    CK(check(a));
    FT_CHECK_ARG(dctx && dQ && dV && dtext && dw_key && dv && dw1 && db1 && dw2 && db2);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (fused_supported(a))
        return a->mode == FT_F16 ? ftint_cummf_bwd_f16(a, dctx, dattn, dlogprob, dQ, dV, dtext, dw_key, dv, dw1, db1, dw2, db2, st)
                                 : ftint_cummf_bwd(a, dctx, dattn, dlogprob, dQ, dV, dtext, dw_key, dv, dw1, db1, dw2, db2, st);
    const int T = a->T, B = a->B, L = a->L, E = a->E, A = a->A, R = L * B, NF = a->NF, C1 = 2 * a->K1, C2n = NF * a->K2;
    const Buf b = carve(a->work, L, B, E, A, NF, a->K1, a->K2, true);
    FT_CHECK_ARG(a->work_bytes >= b.total);
    hipLaunchKernelGGL(fill_int_k, dim3(cdiv(B, 256)), dim3(256), 0, st, b.full_lens, B, L);
    // every accumulated output starts from zero (the caller hands over uninitialised buffers)
    FT_CHECK_HIP(hipMemsetAsync(dV, 0, sizeof(float) * (size_t)R * A, st));
    FT_CHECK_HIP(hipMemsetAsync(dtext, 0, sizeof(float) * (size_t)R * E, st));
    FT_CHECK_HIP(hipMemsetAsync(dw_key, 0, sizeof(float) * (size_t)A * E, st));
    FT_CHECK_HIP(hipMemsetAsync(dv, 0, sizeof(float) * (size_t)A, st));
    FT_CHECK_HIP(hipMemsetAsync(dw1, 0, sizeof(float) * (size_t)NF * C1, st));
    FT_CHECK_HIP(hipMemsetAsync(db1, 0, sizeof(float) * (size_t)NF, st));
    FT_CHECK_HIP(hipMemsetAsync(dw2, 0, sizeof(float) * (size_t)E * C2n, st));
    FT_CHECK_HIP(hipMemsetAsync(db2, 0, sizeof(float) * (size_t)E, st));
    FT_CHECK_HIP(hipMemsetAsync(b.g_prev, 0, sizeof(float) * (size_t)B * L, st));
    FT_CHECK_HIP(hipMemsetAsync(b.g_cumm, 0, sizeof(float) * (size_t)B * L, st));
    const size_t lds = sizeof(float) * ((size_t)2 * ((L + 3) & ~3) + 512);
    const float inv_temp = 1.0f / a->temperature;
    for (int i = T - 1; i >= 0; --i) {
        const float* cumm_i = a->cumm_all + (size_t)i * B * L;
        const float* prev_i = i > 0 ? a->attn + (size_t)(i - 1) * L : nullptr;
        const float* kp = a->kproj_all + (size_t)i * R * A;
        // 1. scores / softmax / context backward: dQ_i, dK_i, dV, dv
        hipLaunchKernelGGL(cumm_dp_k, dim3(cdiv(L, 16), B), dim3(256), 0, st, a->V, a->in_lens, a->attn, dctx, dattn, dlogprob,
                           b.g_prev, b.g_cumm, b.esc, i, T, B, L, A);
        hipLaunchKernelGGL(cumm_score_bwd_k, dim3(cdiv(A, 64), B), dim3(256), lds, st, a->Q, kp, a->v, a->in_lens, a->attn, dctx, b.esc,
                           dQ, b.dkp, dV, dv, i, T, B, L, A, inv_temp);
        // 2. the frame's location features again (cond_i, km_i, h1, col1, col2)
        CK(cond_chain(a, b, cumm_i, prev_i, st));
        // 3. key projection backward: dW_key += dK^T km ; dkm = dK W_key
        CK(gemm(b.dkp, b.km, dw_key, nullptr, A, E, R, 1, A, E, 1, E, 1.f, FT_ACT_NONE, a->mode, FT_GEMM_SPLITK, b, st));
        CK(gemm(b.dkp, a->w_key, b.dkm, nullptr, R, E, A, A, 1, E, 1, E, 0.f, FT_ACT_NONE, a->mode, 0, b, st));
        // 4. km = text . cond:  dtext += dkm . cond ;  dcond = dkm . text ;  through the sigmoid
        hipLaunchKernelGGL(fma_acc_k, dim3(2048), dim3(256), 0, st, b.dkm, b.cond, dtext, (long)R * E);
        CK(ft_eltwise(b.dkm, a->text, b.dcond, (int64_t)R * E, 1, st));
        CK(ft_act_bwd(b.cond, b.dcond, b.dkm, (int64_t)R * E, FT_ACT_SIGMOID, st));                     // dpre2 -> b.dkm
        // 5. second convolution backward
        CK(gemm(b.dkm, b.col2, dw2, nullptr, E, C2n, R, 1, E, C2n, 1, C2n, 1.f, FT_ACT_NONE, a->mode, FT_GEMM_SPLITK, b, st));
        hipLaunchKernelGGL(colsum_acc_k, dim3(cdiv(E, 64), cdiv(R, 256)), dim3(256), 0, st, b.dkm, (long)R, E, db2);
        CK(gemm(b.dkm, a->w2, b.dcol2, nullptr, R, C2n, E, E, 1, C2n, 1, C2n, 0.f, FT_ACT_NONE, a->mode, 0, b, st));
        CK(ft_col2im(b.dcol2, b.dh1, b.full_lens, L, B, NF, a->K2, st));
        float* dpre1 = b.dcond;                                                                          // (free again: [R, NF] fits)
        CK(ft_act_bwd(b.h1, b.dh1, dpre1, (int64_t)R * NF, FT_ACT_RELU, st));
        // 6. first convolution backward
        CK(gemm(dpre1, b.col1, dw1, nullptr, NF, C1, R, 1, NF, C1, 1, C1, 1.f, FT_ACT_NONE, a->mode, FT_GEMM_SPLITK, b, st));
        hipLaunchKernelGGL(colsum_acc_k, dim3(cdiv(NF, 64), cdiv(R, 256)), dim3(256), 0, st, dpre1, (long)R, NF, db1);
        CK(gemm(dpre1, a->w1, b.dcol1, nullptr, R, C1, NF, NF, 1, C1, 1, C1, 0.f, FT_ACT_NONE, a->mode, 0, b, st));
        CK(ft_col2im(b.dcol1, b.ds2, b.full_lens, L, B, 2, a->K1, st));
        // 7. gradients of this frame's (cumm, prev) inputs: prev feeds attn_{i-1} only, cumm feeds every earlier attention
        hipLaunchKernelGGL(unstack_acc_k, dim3(cdiv(R, 256)), dim3(256), 0, st, b.ds2, b.g_prev, b.g_cumm, L, B);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
