// Length-masked LSTM recurrence for training: one kernel launch per time step
// (a kernel boundary is the cheapest chip-wide barrier on gfx950, ~1.5 us, and
// cannot deadlock), each launch spreading the [B,H]x[H,4H] recurrent product
// over H/4 workgroups so every CU streams only its own 16 rows of W_hh (which
// stay resident in its XCD's L2 across steps: blockIdx -> XCD is stable).
//
//   fwd step : block j owns hidden units 4j..4j+3 = 16 gate rows (i,f,g,o x 4).
//              4 waves split K=H; skinny MFMA tiles (M = batch rows, N = 16 rows);
//              cross-wave reduce in LDS; the i/f/g/o nonlinearity, cell update,
//              length masking and all stores are fused in the same launch.
//   bwd step : pointwise kernel (dgates from dh, dc) + split-K skinny MFMA
//              dh_rec = dgates . W_hh (against a once-transposed W_hh^T).
// Weight gradients and dx are plain GEMMs over all T*B rows afterwards.
#include "common.h"

namespace {

// ---- skinny  D[b][n] += sum_k A[b][k] * W[n][k]  over k-chunks c = c0, c0+cs, ... ----
// fp32 : chunk = 16 k, lane (li,kg) loads float4 at k = c*16 + kg*4, 4 x mfma_16x16x4f32
// bf16 : chunk = 32 k, lane loads 8 floats at k = c*32 + kg*8, 1 x mfma_16x16x32_bf16
template <int MODE, int MT>
__device__ __forceinline__ void skinny_nt(const float* const (&arow)[MT], const float* wrow, int K,
                                          int c0, int cs, int kg, f32x4 (&acc)[MT]) {
    constexpr int KC = (MODE == 0) ? 16 : 32;
    const int nchunk = (K + KC - 1) / KC;
    for (int c = c0; c < nchunk; c += cs) {
        if constexpr (MODE == 0) {
            const int k = c * 16 + kg * 4;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wrow && k < K) w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (arow[m] && k < K) a = *reinterpret_cast<const float4*>(arow[m] + k);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[m], 0, 0, 0);
            }
        } else {
            const int k = c * 32 + kg * 8;
            float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
            if (wrow && k < K) w0 = *reinterpret_cast<const float4*>(wrow + k);
            if (wrow && k + 4 < K) w1 = *reinterpret_cast<const float4*>(wrow + k + 4);
            bf16x8 wb;
            wb[0] = f2bf(w0.x); wb[1] = f2bf(w0.y); wb[2] = f2bf(w0.z); wb[3] = f2bf(w0.w);
            wb[4] = f2bf(w1.x); wb[5] = f2bf(w1.y); wb[6] = f2bf(w1.z); wb[7] = f2bf(w1.w);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
                if (arow[m] && k < K) a0 = *reinterpret_cast<const float4*>(arow[m] + k);
                if (arow[m] && k + 4 < K) a1 = *reinterpret_cast<const float4*>(arow[m] + k + 4);
                bf16x8 ab;
                ab[0] = f2bf(a0.x); ab[1] = f2bf(a0.y); ab[2] = f2bf(a0.z); ab[3] = f2bf(a0.w);
                ab[4] = f2bf(a1.x); ab[5] = f2bf(a1.y); ab[6] = f2bf(a1.z); ab[7] = f2bf(a1.w);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, wb, acc[m], 0, 0, 0);
            }
        }
    }
}

struct FwdP {
    const float* gx; const float* w_hh; const int* lens;
    const float* hprev; float* hnext; float* cstate;
    float* y; long ldy; float* gates; float* cell;
    int s, T, B, H, reverse;
};

template <int MODE, int MT>
__global__ __launch_bounds__(256) void lstm_fwd_step(FwdP p) {
    __shared__ float red[4][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int u0 = blockIdx.x * 4;
    const int H = p.H, B = p.B;

    // B operand row for this lane: n = li = gate*4 + ul  ->  W_hh row gate*H + u0 + ul
    const int wu = u0 + (li & 3);
    const float* wrow = (wu < H) ? p.w_hh + (size_t)((li >> 2) * H + wu) * H : nullptr;
    const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int b = m * 16 + li;
        arow[m] = (b < B) ? p.hprev + (size_t)b * H : nullptr;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    skinny_nt<MODE, MT>(arow, wrow, H, wave, 4, kg, acc);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + kg * 4 + r][li] = acc[m][r];
    __syncthreads();

    const int b = tid >> 2, ul = tid & 3, u = u0 + ul;
    if (b >= B || u >= H) return;
    const int len = p.lens[b];
    const bool active = p.s < len;
    const size_t bu = (size_t)b * H + u;
    if (!active) {                               // sample finished: pad row of y is zero, state frozen
        p.y[((size_t)p.s * B + b) * p.ldy + u] = 0.f;
        p.hnext[bu] = p.hprev[bu];
        return;
    }
    const int t = p.reverse ? (len - 1 - p.s) : p.s;
    const size_t row = (size_t)t * B + b;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int n = g * 4 + ul;
        pre[g] = red[0][b][n] + red[1][b][n] + red[2][b][n] + red[3][b][n] + p.gx[row * 4 * H + (size_t)g * H + u];
    }
    const float ig = 1.f / (1.f + expf(-pre[0]));
    const float fg = 1.f / (1.f + expf(-pre[1]));
    const float gg = tanhf(pre[2]);
    const float og = 1.f / (1.f + expf(-pre[3]));
    const float c_new = fg * p.cstate[bu] + ig * gg;
    const float h_new = og * tanhf(c_new);
    p.cstate[bu] = c_new;
    p.hnext[bu] = h_new;
    p.y[row * p.ldy + u] = h_new;
    if (p.gates) {
        float* gp = p.gates + row * 4 * H + u;
        gp[0] = ig; gp[(size_t)H] = fg; gp[(size_t)2 * H] = gg; gp[(size_t)3 * H] = og;
        p.cell[row * H + u] = c_new;
    }
}

struct BwdP {
    const float* dy; long ldy; const int* lens;
    const float* gates; const float* cell;
    const float* part;      // [4][B][H] partial dh_rec from step s+1
    float* dc_carry;        // [B][H]
    float* da_cur;          // [B][4H]
    float* dgx;             // [T][B][4H]
    const float* wT;        // [H][4H]
    float* part_out;        // [4][B][H]
    int s, T, B, H, reverse;
};

__global__ void lstm_bwd_pointwise(BwdP p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int H = p.H, B = p.B;
    if (idx >= B * H) return;
    const int b = idx / H, u = idx - b * H;
    const int len = p.lens[b];
    const bool active = p.s < len;
    const size_t BH = (size_t)B * H;
    float da[4] = {0.f, 0.f, 0.f, 0.f};
    int t = p.s;
    if (active) {
        t = p.reverse ? (len - 1 - p.s) : p.s;
        const size_t row = (size_t)t * B + b;
        float dh = p.dy[row * p.ldy + u] + p.part[idx] + p.part[BH + idx] + p.part[2 * BH + idx] + p.part[3 * BH + idx];
        const float* gp = p.gates + row * 4 * H + u;
        const float ig = gp[0], fg = gp[(size_t)H], gg = gp[(size_t)2 * H], og = gp[(size_t)3 * H];
        const float c_t = p.cell[row * H + u];
        float c_prev = 0.f;
        if (p.s > 0) {
            const int tp = p.reverse ? t + 1 : t - 1;
            c_prev = p.cell[((size_t)tp * B + b) * H + u];
        }
        const float tc = tanhf(c_t);
        const float dc = dh * og * (1.f - tc * tc) + p.dc_carry[idx];
        p.dc_carry[idx] = dc * fg;
        da[0] = dc * gg * ig * (1.f - ig);
        da[1] = dc * c_prev * fg * (1.f - fg);
        da[2] = dc * ig * (1.f - gg * gg);
        da[3] = dh * tc * og * (1.f - og);
    }
    float* dg = p.dgx + ((size_t)t * B + b) * 4 * H + u;       // inactive: t == s is a pad row -> zeros
    float* dc_ = p.da_cur + (size_t)b * 4 * H + u;
#pragma unroll
    for (int g = 0; g < 4; ++g) { dg[(size_t)g * H] = da[g]; dc_[(size_t)g * H] = da[g]; }
}

// part_out[ks][b][j] = sum_{r in K-slice ks} da_cur[b][r] * wT[j][r]
template <int MODE, int MT>
__global__ __launch_bounds__(256) void lstm_bwd_matmul(BwdP p) {
    __shared__ float red[4][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B, K4 = 4 * p.H;
    const int j0 = blockIdx.x * 16, ks = blockIdx.y;
    const int KR = K4 / 4;                         // K-slice length (= H), multiple of 4
    const int j = j0 + li;
    const float* wrow = (j < H) ? p.wT + (size_t)j * K4 + (size_t)ks * KR : nullptr;
    const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int b = m * 16 + li;
        arow[m] = (b < B) ? p.da_cur + (size_t)b * K4 + (size_t)ks * KR : nullptr;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    skinny_nt<MODE, MT>(arow, wrow, KR, wave, 4, kg, acc);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + kg * 4 + r][li] = acc[m][r];
    __syncthreads();
    for (int i = tid; i < MT * 16 * 16; i += 256) {
        int b = i >> 4, n = i & 15;
        if (b < B && j0 + n < H)
            p.part_out[((size_t)ks * B + b) * H + j0 + n] = red[0][b][n] + red[1][b][n] + red[2][b][n] + red[3][b][n];
    }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Ccols) {
    __shared__ float tile[32][33];
    int c = blockIdx.x * 32 + threadIdx.x;
    for (int i = threadIdx.y; i < 32; i += 8) {
        int r = blockIdx.y * 32 + i;
        if (r < R && c < Ccols) tile[i][threadIdx.x] = in[(size_t)r * Ccols + c];
    }
    __syncthreads();
    int r2 = blockIdx.y * 32 + threadIdx.x;
    for (int i = threadIdx.y; i < 32; i += 8) {
        int c2 = blockIdx.x * 32 + i;
        if (r2 < R && c2 < Ccols) out[(size_t)c2 * R + r2] = tile[threadIdx.x][i];
    }
}

template <int MODE>
void launch_fwd(const FwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm_fwd_step<MODE, 1>), grid, dim3(256), 0, st, p);
    else if (mt == 2) hipLaunchKernelGGL((lstm_fwd_step<MODE, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm_fwd_step<MODE, 4>), grid, dim3(256), 0, st, p);
}
template <int MODE>
void launch_bwd_mm(const BwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm_bwd_matmul<MODE, 1>), grid, dim3(256), 0, st, p);
    else if (mt == 2) hipLaunchKernelGGL((lstm_bwd_matmul<MODE, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm_bwd_matmul<MODE, 4>), grid, dim3(256), 0, st, p);
}

}  // namespace

extern "C" size_t ft_lstm_workspace_bytes(int B, int H) {
    size_t fwd = (size_t)3 * B * H;
    size_t bwd = (size_t)B * 4 * H + (size_t)4 * B * H + (size_t)B * H + (size_t)4 * H * H;
    return sizeof(float) * (fwd > bwd ? fwd : bwd);
}

extern "C" int ft_lstm_seq_fwd(const float* gx, const float* w_hh, const int32_t* lens,
                               float* y, int64_t ldy, float* gates, float* cell, void* work,
                               int T, int B, int H, int reverse, int mode, void* stream) {
    FT_CHECK_ARG(gx && w_hh && lens && y && work);
    FT_CHECK_ARG(T >= 0 && B >= 1 && B <= 64 && H >= 4 && H % 4 == 0 && ldy >= H);
    FT_CHECK_ARG((gates == nullptr) == (cell == nullptr));
    FT_CHECK_ARG(mode == FT_F32 || mode == FT_BF16);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(w_hh) % 16 == 0 && reinterpret_cast<uintptr_t>(work) % 16 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* w = reinterpret_cast<float*>(work);
    const size_t BH = (size_t)B * H;
    float* hbuf[2] = {w, w + BH};
    float* cstate = w + 2 * BH;
    FT_CHECK_HIP(hipMemsetAsync(w, 0, 3 * BH * sizeof(float), st));
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    dim3 grid(cdiv(H, 4));
    for (int s = 0; s < T; ++s) {
        FwdP p{gx, w_hh, lens, hbuf[s & 1], hbuf[(s + 1) & 1], cstate, y, (long)ldy, gates, cell, s, T, B, H, reverse};
        if (mode == FT_F32) launch_fwd<0>(p, mt, grid, st);
        else launch_fwd<1>(p, mt, grid, st);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int ft_lstm_seq_bwd(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens,
                               const float* gates, const float* cell, float* dgx, void* work,
                               int T, int B, int H, int reverse, int mode, void* stream) {
    FT_CHECK_ARG(dy && w_hh && lens && gates && cell && dgx && work);
    FT_CHECK_ARG(T >= 0 && B >= 1 && B <= 64 && H >= 4 && H % 4 == 0 && ldy >= H);
    FT_CHECK_ARG(mode == FT_F32 || mode == FT_BF16);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(work) % 16 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* w = reinterpret_cast<float*>(work);
    const size_t BH = (size_t)B * H;
    float* da_cur = w;                 // [B][4H]
    float* part = w + 4 * BH;          // [4][B][H]
    float* dc_carry = w + 8 * BH;      // [B][H]
    float* wT = w + 9 * BH;            // [H][4H]
    FT_CHECK_HIP(hipMemsetAsync(w, 0, 9 * BH * sizeof(float), st));
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(H, 32), cdiv(4 * H, 32)), dim3(32, 8), 0, st, w_hh, wT, 4 * H, H);
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    dim3 grid_pw(cdiv((int64_t)B * H, 256)), grid_mm(cdiv(H, 16), 4);
    for (int s = T - 1; s >= 0; --s) {
        BwdP p{dy, (long)ldy, lens, gates, cell, part, dc_carry, da_cur, dgx, wT, part, s, T, B, H, reverse};
        hipLaunchKernelGGL(lstm_bwd_pointwise, grid_pw, dim3(256), 0, st, p);
        if (s > 0) {
            if (mode == FT_F32) launch_bwd_mm<0>(p, mt, grid_mm, st);
            else launch_bwd_mm<1>(p, mt, grid_mm, st);
        }
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
