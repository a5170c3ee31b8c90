// Autoregressive decode of one flow at batch 1 (reference AR_Step.infer, flowtron.py:775-828).
//
// Batch-1 decode is a chain of dependent GEMVs over 26.8 M weights per frame (107 MB fp32):
// it is bound by streaming those weights (they stay resident in the 256 MiB Infinity Cache
// across frames) and by the eight dependent stages per frame.  Each stage is one launch that
// fills the chip with one wave per output row (or per hidden unit: the four i/f/g/o gate rows
// of a unit are reduced by the same wave so the cell update fuses into the GEMV):
//
//   S1 attention_lstm step      S2 query projection      S3a scores (wave per text position)  S3b softmax + context
//   S4 lstm layer 0 step        S5 lstm layer 1 step     S6/S7 dense tanh x2
//   S8a 1x1 conv -> (log_s,b)   S8b inverse coupling, gate sigmoid/threshold, frame counter
//
// Nothing returns to the host inside the loop: the frame index and the stop flag live in
// device memory (the reference's python `if sigmoid(gate) > thr: break`, flowtron.py:823-826,
// becomes a device flag every stage tests).  All kernels take the same parameter block BY VALUE
// (kernarg segment: weight addresses are known at wave start, so the weight stream is in flight
// before the frame counter has even been read), and a chunk of frames is captured ONCE into a
// hipGraph that is replayed for every chunk of every utterance that reuses the same buffers.
#include <deque>
#include <mutex>
#include <unordered_map>

#include "common.h"
#include <type_traits>

namespace {

typedef unsigned short bf16_t;

struct DecodeDev {
    const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;
    const float *w_query, *v, *K, *V;
    const float *l0_w_ih, *l0_w_hh, *l0_b_ih, *l0_b_hh, *l1_w_ih, *l1_w_hh, *l1_b_ih, *l1_b_hh;
    const float *d0_w, *d0_b, *d1_w, *d1_b, *conv_w, *conv_b, *gate_w, *gate_b;
    const float* residual; float* mel_out; float* attn_out; int* n_done_dev;
    float *h_att, *c_att, *h0, *c0, *h1, *c1;   // h_*: [2][H] ping-pong by frame parity
    float *q, *ctx, *u1, *u2, *prev;
    // cumulative (location-sensitive) attention, flowtron.py:129-152, :793-806 -- all null when use_cumm_attention is off
    const float *cond_w1, *cond_b1, *cond_w2, *cond_b2, *w_key, *enc;
    const float *prior, *forced;                 // [N,L] attention prior (posterior, flowtron.py:544-557) / forced alignment (:585-588)
    float *cumm, *prev_attn, *keyin, *Kdyn;
    float *escore, *obuf;                        // attention scores [L], 1x1 conv output [2M] (stage hand-offs)
    int* ctl;                                    // [0] frame index, [1] done flag
    // bf16 images of the weight matrices (null = stream the fp32 originals)
    const bf16_t *att_w_ih16, *att_w_hh16, *w_query16, *l0_w_ih16, *l0_w_hh16, *l1_w_ih16, *l1_w_hh16, *d0_w16, *d1_w16, *conv_w16;
    int N, L, H, A, M, E;
    float inv_temp, gate_threshold;
};

// ---- bf16 weight images (bf16 operand mode): every weight matrix of the flow is rounded ONCE per ft_decode_flow call into a
// bf16 copy (53.7 MB instead of 107.4 MB per frame and flow; the copy stays in the Infinity Cache across frames) and the
// GEMVs stream those; activations and accumulation stay fp32.  16-byte loads = 8 weights per lane.
__device__ __forceinline__ float dot8(const uint4 w, const float4 xa, const float4 xb) {
    return __uint_as_float(w.x << 16) * xa.x + __uint_as_float(w.x & 0xffff0000u) * xa.y + __uint_as_float(w.y << 16) * xa.z +
           __uint_as_float(w.y & 0xffff0000u) * xa.w + __uint_as_float(w.z << 16) * xb.x + __uint_as_float(w.z & 0xffff0000u) * xb.y +
           __uint_as_float(w.w << 16) * xb.z + __uint_as_float(w.w & 0xffff0000u) * xb.w;
}
// one weight row against an fp32 activation segment; K % 8 == 0, 16-byte aligned (checked by the host for the bf16 path)
__device__ __forceinline__ float dot_seg(const bf16_t* __restrict__ w, const float* __restrict__ x, int K, int lane) {
    const uint4* w8 = reinterpret_cast<const uint4*>(w);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int K8 = K >> 3;
    float s = 0.f;
    for (int kb = 0; kb < K8; kb += 256) {                       // 4 loads per lane in flight
        uint4 wv[4];
        float4 xa[4], xb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + u * 64 + lane;
            const bool ok = k < K8;
            const int kk = ok ? k : 0;
            wv[u] = ok ? w8[kk] : make_uint4(0u, 0u, 0u, 0u);
            xa[u] = x4[2 * kk]; xb[u] = x4[2 * kk + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += dot8(wv[u], xa[u], xb[u]);
    }
    return s;
}

__device__ __forceinline__ float dot_seg(const float* __restrict__ w, const float* __restrict__ x, int K, int lane) {
    float s = 0.f;
    if (((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
        const float4* w4 = reinterpret_cast<const float4*>(w);
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int K4 = K >> 2;
        for (int k = lane; k < K4; k += 64) {
            const float4 a = w4[k], b = x4[k];
            s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
    } else {
        for (int k = lane; k < K; k += 64) s += w[k] * x[k];
    }
    return s;
}

// Four weight rows (stride gstride floats apart) against one activation segment, all U*5 16-byte loads of a pass in
// flight before the FMAs (a decode GEMV is a pure weight stream: bytes in flight per CU decide the rate).
// Requires K % 4 == 0 and 16-byte aligned w / x.  Out-of-range lanes read element 0 against a zero activation.
template <int U>
__device__ __forceinline__ void dot4_seg(const float* __restrict__ w, size_t gstride, const float* __restrict__ x, int K,
                                         int lane, float (&acc)[4]) {
    const int K4 = K >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int kb = 0; kb < K4; kb += 64 * U) {
        float4 xv[U], wv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = kb + u * 64 + lane;
            const bool ok = k < K4;
            const int kk = ok ? k : 0;
            xv[u] = x4[kk];
            if (!ok) xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int g = 0; g < 4; ++g) wv[u][g] = reinterpret_cast<const float4*>(w + (size_t)g * gstride)[kk];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[g] += wv[u][g].x * xv[u].x + wv[u][g].y * xv[u].y + wv[u][g].z * xv[u].z + wv[u][g].w * xv[u].w;
    }
}

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ bool frame_live(const DecodeDev& P, int& i) {
    i = P.ctl[0];
    return (P.ctl[1] == 0) && (i < P.N);
}

// One wave per hidden unit u: gates_g = W_ih[g*H+u,:].x (+ second input segment) + W_hh[g*H+u,:].h + b
template <int WHICH>   // 0 attention_lstm, 1 lstm l0, 2 lstm l1
__global__ __launch_bounds__(256) void dec_lstm_k(const DecodeDev P) {
    int i;
    const bool live = frame_live(P, i);        // the exit is taken AFTER the GEMV: weight addresses do not depend on the
    const int lane = threadIdx.x & 63;         // frame counter, so their loads are in flight while ctl is still on its way
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = P.H;
    if (u >= H) return;
    const int par = i & 1;
    const float *w_ih, *w_hh, *b_ih, *b_hh, *x0, *x1 = nullptr;
    float *hbuf, *cbuf;
    int K0, K1 = 0;
    if (WHICH == 0) {
        w_ih = P.att_w_ih; w_hh = P.att_w_hh; b_ih = P.att_b_ih; b_hh = P.att_b_hh;
        x0 = P.prev; K0 = P.M; hbuf = P.h_att; cbuf = P.c_att;
    } else if (WHICH == 1) {
        w_ih = P.l0_w_ih; w_hh = P.l0_w_hh; b_ih = P.l0_b_ih; b_hh = P.l0_b_hh;
        x0 = P.h_att + (par ^ 1) * H; K0 = H; x1 = P.ctx; K1 = P.A; hbuf = P.h0; cbuf = P.c0;
    } else {
        w_ih = P.l1_w_ih; w_hh = P.l1_w_hh; b_ih = P.l1_b_ih; b_hh = P.l1_b_hh;
        x0 = P.h0 + (par ^ 1) * H; K0 = H; hbuf = P.h1; cbuf = P.c1;
    }
    const float* hold = hbuf + par * H;
    const int Kin = K0 + K1;
    float pre[4];
    const bool vec = (((Kin | K0 | K1 | H) & 3) == 0) && aligned16(w_ih) && aligned16(w_hh) && aligned16(x0) &&
                     aligned16(hold) && (K1 == 0 || aligned16(x1));
    if (vec) {           // the four gate rows of unit u stream together
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        dot4_seg<4>(w_ih + (size_t)u * Kin, (size_t)H * Kin, x0, K0, lane, acc);
        if (K1) dot4_seg<4>(w_ih + (size_t)u * Kin + K0, (size_t)H * Kin, x1, K1, lane, acc);
        dot4_seg<4>(w_hh + (size_t)u * H, (size_t)H * H, hold, H, lane, acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t row = (size_t)g * H + u;
            pre[g] = wave_sum(acc[g]) + b_ih[row] + b_hh[row];
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t row = (size_t)g * H + u;
            float s = dot_seg(w_ih + row * Kin, x0, K0, lane);
            if (K1) s += dot_seg(w_ih + row * Kin + K0, x1, K1, lane);
            s += dot_seg(w_hh + row * H, hold, H, lane);
            pre[g] = wave_sum(s) + b_ih[row] + b_hh[row];
        }
    }
    if (live && lane == 0) {
        const float ig = 1.f / (1.f + expf(-pre[0]));
        const float fg = 1.f / (1.f + expf(-pre[1]));
        const float gg = tanhf(pre[2]);
        const float og = 1.f / (1.f + expf(-pre[3]));
        const float c = fg * cbuf[u] + ig * gg;
        cbuf[u] = c;
        hbuf[(par ^ 1) * H + u] = og * tanhf(c);
    }
}

// bf16-weight form: ONE WORKGROUP PER HIDDEN UNIT, wave g streams gate row g (all of its 16-byte loads in flight): H workgroups
// = 4 per CU = 16 waves per CU pulling weights, instead of 4 -- a decode GEMV is bound by the bytes a CU has in flight.
template <int WHICH>
__global__ __launch_bounds__(256) void dec_lstm16_k(const DecodeDev P) {
    __shared__ float pre_s[4];
    int i;
    const bool live = frame_live(P, i);
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int u = blockIdx.x, H = P.H;
    const int par = i & 1;
    const bf16_t *w_ih, *w_hh;
    const float *b_ih, *b_hh, *x0, *x1 = nullptr;
    float *hbuf, *cbuf;
    int K0, K1 = 0;
    if (WHICH == 0) {
        w_ih = P.att_w_ih16; w_hh = P.att_w_hh16; b_ih = P.att_b_ih; b_hh = P.att_b_hh;
        x0 = P.prev; K0 = P.M; hbuf = P.h_att; cbuf = P.c_att;
    } else if (WHICH == 1) {
        w_ih = P.l0_w_ih16; w_hh = P.l0_w_hh16; b_ih = P.l0_b_ih; b_hh = P.l0_b_hh;
        x0 = P.h_att + (par ^ 1) * H; K0 = H; x1 = P.ctx; K1 = P.A; hbuf = P.h0; cbuf = P.c0;
    } else {
        w_ih = P.l1_w_ih16; w_hh = P.l1_w_hh16; b_ih = P.l1_b_ih; b_hh = P.l1_b_hh;
        x0 = P.h0 + (par ^ 1) * H; K0 = H; hbuf = P.h1; cbuf = P.c1;
    }
    const float* hold = hbuf + par * H;
    const int Kin = K0 + K1;
    const size_t row = (size_t)g * H + u;
    float s = dot_seg(w_ih + row * Kin, x0, K0, lane);
    if (K1) s += dot_seg(w_ih + row * Kin + K0, x1, K1, lane);
    s += dot_seg(w_hh + row * H, hold, H, lane);
    s = wave_sum(s);
    if (lane == 0) pre_s[g] = s + b_ih[row] + b_hh[row];
    __syncthreads();
    if (live && threadIdx.x == 0) {
        const float ig = 1.f / (1.f + expf(-pre_s[0]));
        const float fg = 1.f / (1.f + expf(-pre_s[1]));
        const float gg = tanhf(pre_s[2]);
        const float og = 1.f / (1.f + expf(-pre_s[3]));
        const float c = fg * cbuf[u] + ig * gg;
        cbuf[u] = c;
        hbuf[(par ^ 1) * H + u] = og * tanhf(c);
    }
}

// y[n] = act(W[n,:].x + b[n]), one wave per row.  WHICH: 0 query (x = new h_att), 1 dense0 (x = new h1), 2 dense1 (x = u1)
template <int WHICH>
__global__ __launch_bounds__(256) void dec_gemv_k(const DecodeDev P) {
    int i;
    const bool live = frame_live(P, i);
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = P.H;
    const int par = i & 1;
    const float *W, *b = nullptr, *x;
    float* y;
    int N, K;
    if (WHICH == 0) { W = P.w_query; x = P.h_att + (par ^ 1) * H; y = P.q; N = P.A; K = H; }
    else if (WHICH == 1) { W = P.d0_w; b = P.d0_b; x = P.h1 + (par ^ 1) * H; y = P.u1; N = H; K = H; }
    else { W = P.d1_w; b = P.d1_b; x = P.u1; y = P.u2; N = H; K = H; }
    if (n >= N) return;
    const bf16_t* W16 = WHICH == 0 ? P.w_query16 : (WHICH == 1 ? P.d0_w16 : P.d1_w16);
    float s = wave_sum(W16 ? dot_seg(W16 + (size_t)n * K, x, K, lane) : dot_seg(W + (size_t)n * K, x, K, lane));
    if (live && lane == 0) {
        if (WHICH != 0) s = tanhf(s + b[n]);
        y[n] = s;
    }
}

// S3a  attention scores: one wave per text position l (grid = ceil(L/4)); e[l] = (1/temp) sum_a v[a] tanh(q[a] + K[l][a])
__global__ __launch_bounds__(256) void dec_score_k(const DecodeDev P) {
    int i;
    if (!frame_live(P, i)) return;
    const int L = P.L, A = P.A;
    const int lane = threadIdx.x & 63, l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= L || P.forced) return;             // forced alignment: the scores are not used
    const float* Kmat = P.Kdyn ? P.Kdyn : P.K;
    const float* kr = Kmat + (size_t)l * A;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += P.v[a] * tanhf(P.q[a] + kr[a]);
    s = wave_sum(s);
    if (lane == 0) P.escore[l] = s * P.inv_temp;
}

// S3b  softmax over L (recomputed by every workgroup: L floats) + context for 64 channels per workgroup (grid = ceil(A/64));
//      workgroup 0 also stores the attention row and advances the cumulative-attention state.
__global__ __launch_bounds__(256) void dec_ctx_k(const DecodeDev P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [L] probabilities, [4] reduction, [4][64] partial context
    int i;
    if (!frame_live(P, i)) return;
    const int L = P.L, A = P.A;
    float* e = sm;
    float* red = sm + L;
    float* part = red + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto softmax_inplace = [&]() {              // e[0..L) <- softmax(e), all 256 threads
        float m = -INFINITY;
        for (int l = tid; l < L; l += 256) m = fmaxf(m, e[l]);
        m = wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float s = 0.f;
        for (int l = tid; l < L; l += 256) { const float p = expf(e[l] - m); e[l] = p; s += p; }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        s = red[0] + red[1] + red[2] + red[3];
        for (int l = tid; l < L; l += 256) e[l] = e[l] / s;
        __syncthreads();
    };
    if (P.forced) {
        for (int l = tid; l < L; l += 256) e[l] = P.forced[(size_t)i * L + l];
        __syncthreads();
    } else {
        for (int l = tid; l < L; l += 256) e[l] = P.escore[l];
        __syncthreads();
        softmax_inplace();
        if (P.prior) {                            // posterior with the prior row of this frame, then a second softmax
            for (int l = tid; l < L; l += 256) e[l] = logf(e[l] + 1e-20f) + logf(P.prior[(size_t)i * L + l] + 1e-20f);
            __syncthreads();
            softmax_inplace();
        }
    }
    if (blockIdx.x == 0) {
        float* arow = P.attn_out + (size_t)i * L;
        for (int l = tid; l < L; l += 256) {
            const float pl = e[l];
            arow[l] = pl;
            if (P.cumm) { P.prev_attn[l] = pl; P.cumm[l] += pl; }      // read by the NEXT frame's dec_cond_k
        }
    }
    const int a = blockIdx.x * 64 + lane;
    float c = 0.f;
    if (a < A)
        for (int l = wave; l < L; l += 4) c += e[l] * P.V[(size_t)l * A + a];
    part[wave * 64 + lane] = c;
    __syncthreads();
    if (wave == 0 && a < A) P.ctx[a] = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
}

// Location features -> key modulation for text position l (one workgroup per l):
//   h1[l'][c] = relu(b1[c] + sum_{ch<2,k<5} w1[c][ch][k] * x_ch[l'+k-2]),  x_0 = cumulative attention, x_1 = previous attention
//   cond[l][e] = sigmoid(b2[e] + sum_{c<32,k<3} w2[e][c][k] * h1[l+k-1][c]);   keyin[l][e] = enc[l][e] * cond[l][e]
__global__ __launch_bounds__(256) void dec_cond_k(const DecodeDev P) {
    __shared__ float h1[3][32];
    int i;
    if (!frame_live(P, i)) return;
    const int L = P.L, E = P.E, l = blockIdx.x, tid = threadIdx.x;
    if (tid < 96) {
        const int j = tid >> 5, c = tid & 31, lp = l + j - 1;
        float v = 0.f;
        if (lp >= 0 && lp < L) {
            v = P.cond_b1[c];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int ls = lp + k - 2;
                if (ls >= 0 && ls < L) v += P.cond_w1[(c * 2 + 0) * 5 + k] * P.cumm[ls] + P.cond_w1[(c * 2 + 1) * 5 + k] * P.prev_attn[ls];
            }
            v = fmaxf(v, 0.f);
        }
        h1[j][c] = v;                      // zero outside [0,L): Conv1d zero padding of the second conv
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        float v = P.cond_b2[e];
        const float* w = P.cond_w2 + (size_t)e * 96;
#pragma unroll
        for (int c = 0; c < 32; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) v += w[c * 3 + k] * h1[k][c];
        const float cond = 1.f / (1.f + expf(-v));
        P.keyin[(size_t)l * E + e] = P.enc[(size_t)l * E + e] * cond;
    }
}

// Kdyn[l][a] = sum_e w_key[a][e] * keyin[l][e]; grid (L, ceil(A/16)), one wave per 4 rows a
__global__ __launch_bounds__(256) void dec_key_k(const DecodeDev P) {
    int i;
    if (!frame_live(P, i)) return;
    const int A = P.A, E = P.E, l = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = P.keyin + (size_t)l * E;
    for (int r = 0; r < 4; ++r) {
        const int a = blockIdx.y * 16 + wave * 4 + r;
        if (a >= A) break;
        const float s = wave_sum(dot_seg(P.w_key + (size_t)a * E, x, E, lane));
        if (lane == 0) P.Kdyn[(size_t)l * A + a] = s;
    }
}

// S8a  1x1 conv: one wave per output row n < 2M (grid = ceil(2M/4)); obuf[n] = conv_w[n,:].u2 + conv_b[n]
__global__ __launch_bounds__(256) void dec_conv_k(const DecodeDev P) {
    int i;
    if (!frame_live(P, i)) return;
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= 2 * P.M) return;
    const float s = wave_sum(P.conv_w16 ? dot_seg(P.conv_w16 + (size_t)n * P.H, P.u2, P.H, lane)
                                         : dot_seg(P.conv_w + (size_t)n * P.H, P.u2, P.H, lane));
    if (lane == 0) P.obuf[n] = s + P.conv_b[n];
}

// S8b  inverse affine coupling, gate sigmoid/threshold (flowtron.py:823-826), frame bookkeeping -- one small workgroup
__global__ __launch_bounds__(256) void dec_fin_k(const DecodeDev P) {
    __shared__ float red[4];
    int i;
    if (!frame_live(P, i)) return;
    const int M = P.M, H = P.H, A = P.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int par = i & 1;
    float g = 0.f;
    if (P.gate_w) {
        const float* hn = P.h_att + (par ^ 1) * H;
        for (int k = tid; k < H; k += 256) g += P.gate_w[k] * hn[k];
        for (int k = tid; k < A; k += 256) g += P.gate_w[H + k] * P.ctx[k];
        g = wave_sum(g);
        if (lane == 0) red[wave] = g;
    }
    __syncthreads();
    for (int c = tid; c < M; c += 256) {
        const float x = (P.residual[(size_t)i * M + c] - P.obuf[M + c]) / expf(P.obuf[c]);
        P.mel_out[(size_t)i * M + c] = x;
        P.prev[c] = x;
    }
    if (tid == 0) {
        int done = 0;
        if (P.gate_w) {
            const float gs = P.gate_b[0] + red[0] + red[1] + red[2] + red[3];
            const float sg = 1.f / (1.f + expf(-gs));
            if (sg > P.gate_threshold) done = 1;
        }
        P.ctl[0] = i + 1;
        P.n_done_dev[0] = i + 1;
        if (done) P.ctl[1] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent decode: ONE launch per flow (bf16 weight images, no cumulative attention / prior / forced alignment).
// The staged chain above spends most of a frame in fixed costs -- ten kernel boundaries and, in every stage, a dependent
// chain frame counter -> pointers -> data (~4 us per stage, ~52 us per frame and flow) -- not in streaming weights.  Here 256
// workgroups (one per CU) walk the frames themselves; every stage's output vector is handed to all workgroups as 8-byte
// {epoch, fp32} granules (one write-through store each, tag-checked sc1 loads, no fences -- cdna_hip_programming.md G16 R2,
// ~0.6 us per hop across XCDs) and kept in LDS, and the bf16 weight rows of a stage are REQUESTED BEFORE the wait for its
// input, so the weight stream (L2 / Infinity Cache) hides under the hand-off.
//   workgroup c owns hidden units 4c .. 4c+3 of the three LSTMs (wave w = unit 4c + w, its four gate rows together; the
//   cell states never leave the wave), rows 4c .. 4c+3 of the two dense layers, query rows {c, c+256, c+512}, text
//   position(s) c (+256 ..) of the scores, context channels {c, c+256, c+512} and row c of the 1x1 conv.
// Hops per frame: o -> S1 (inverse coupling of the previous frame + attention LSTM), h_att -> S2 (query), q -> S3a (scores),
// scores -> S3b (softmax + context), ctx -> S4 (gate, LSTM 0), h0 -> S5 (LSTM 1), h1 -> S6, u1 -> S7, u2 -> S8 (conv).
struct DecP {
    DecodeDev d;
    unsigned long long* gran;     // granule buffers, one per stage vector (offsets below, in granules)
    unsigned* census;             // 8 counters behind the nine granule copies (zeroed with them)
    int* status;
    long timeout_ticks;
    long* prof;                   // debug: [frame][12] wall-clock stamps of workgroup 0 (ft_decode_debug_prof), or null
};
enum { G_O = 0, G_HATT = 256, G_Q = 256 + 1024, G_SC = G_Q + 640, G_CTX = G_SC + 1024, G_H0 = G_CTX + 640, G_H1 = G_H0 + 1024,
       G_U1 = G_H1 + 1024, G_U2 = G_U1 + 1024, G_TOTAL = G_U2 + 1024 };

typedef __attribute__((address_space(1))) unsigned long long dgu64;
typedef __attribute__((ext_vector_type(4))) unsigned int du32x4;
constexpr int DEC_LAUX = 2;         // aux bits of the XCD-local gather loads: 2 = nt (as lstm_persist.hip's default), 16 = sc1

__device__ __forceinline__ void publish(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store((dgu64*)g, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// into this XCD's copy: stays in its L2
__device__ __forceinline__ void publish_local(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store((dgu64*)g, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Hand-off topology.  256 workgroups each polling a whole vector across the fabric is 2 MB of sc1 reads per pass -- measured,
// that contention (not the weight stream) is what a stage costs: ~3.5 us per hop against 0.63 us for a lone poller
// (scripts/exp/handoff_probe.hip).  So every vector crosses the fabric ONCE PER XCD: the 32 workgroups of an XCD (run-time census,
// as lstm_persist.hip) each relay a 1/32 slice from the global copy (sc1 polls, 16 lanes) into their XCD's own copy with
// workgroup-scope stores that stay in that XCD's L2, and all of them gather the whole vector from the local copy (~0.25 us).
// Tags travel with the data, so a granule is only ever forwarded / consumed when it shows the epoch: no fences anywhere.
struct Relay {
    unsigned long long* glob;     // p.gran: the producers' copy (write-through stores)
    unsigned long long* loc;      // this XCD's copy
    int q;                        // rank of this workgroup inside its XCD, 0..31
};

// all 256 threads: granules [0, n) of stage vector `off` (epoch-tagged, n <= 1024) -> dst[0, n) in LDS.  false = timed out.
// Granules [0, relay_lo) were produced INSIDE this XCD (stages every XCD computes for itself, below); [relay_lo, n) come from
// the chip-wide producers through the relay (relay_lo even).
__device__ __forceinline__ bool gather(const Relay& R, int off, int n, int relay_lo, unsigned epoch, float* dst, const DecP& p, long t_start) {
    const int npad = (n + 1) & ~1;
    bool ok_all = true;
    if (relay_lo < n) {   // ---- relay: slice q of [relay_lo, n) of the global copy -> local copy; lane pairs of wave 0
        const int S = 2 * ((n - relay_lo + 63) >> 6);
        const int j = relay_lo + R.q * S + 2 * (int)threadIdx.x;
        if ((int)threadIdx.x * 2 < S && j < n) {
            __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(R.glob + off, 0, npad * 8, 0x00020000);
            for (unsigned spins = 0;; ++spins) {
                const du32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rg, j * 8, 0, 16);        // sc1: two granules
                if (v[1] == epoch && (j + 1 >= n || v[3] == epoch)) {
                    __hip_atomic_store((dgu64*)(R.loc + off + j), ((unsigned long long)v[1] << 32) | v[0], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (j + 1 < n)
                        __hip_atomic_store((dgu64*)(R.loc + off + j + 1), ((unsigned long long)v[3] << 32) | v[2], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                if ((spins & 63) == 63 && (wall_clock64() - t_start > p.timeout_ticks ||
                                           __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok_all = false;
                    break;
                }
                asm volatile("" ::: "memory");
            }
        }
    }
    // ---- gather from the XCD-local copy: a thread owns granule pairs 2 tid and 2 tid + 512, re-reads both while stale
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(R.loc + off, 0, npad * 8, 0x00020000);
    const int j0 = threadIdx.x * 2, j1 = j0 + 512;
    bool need0 = j0 < n, need1 = j1 < n;
    for (unsigned spins = 0; ok_all && (need0 | need1); ++spins) {
        du32x4 v0, v1;
        if (need0) v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, j0 * 8, 0, DEC_LAUX);
        if (need1) v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, j1 * 8, 0, DEC_LAUX);
        if (need0 && v0[1] == epoch && (j0 + 1 >= n || v0[3] == epoch)) {
            dst[j0] = __uint_as_float(v0[0]);
            if (j0 + 1 < n) dst[j0 + 1] = __uint_as_float(v0[2]);
            need0 = false;
        }
        if (need1 && v1[1] == epoch && (j1 + 1 >= n || v1[3] == epoch)) {
            dst[j1] = __uint_as_float(v1[0]);
            if (j1 + 1 < n) dst[j1 + 1] = __uint_as_float(v1[2]);
            need1 = false;
        }
        if ((spins & 63) == 63 && (wall_clock64() - t_start > p.timeout_ticks ||
                                   __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            ok_all = false;
            break;
        }
        asm volatile("" ::: "memory");
    }
    if (!ok_all && (threadIdx.x & 63) == 0) atomicExch(p.status, 1);
    return __syncthreads_and(ok_all ? 1 : 0) != 0;
}

// R weight rows (bf16, K % 8 == 0) of one wave: `issue` requests every 16-byte piece (NL per lane and row) -- called BEFORE the
// wait for the stage's input --, `dot` multiplies them with the fp32 activation vector in LDS.
template <int R, int NL>
struct WRows {
    uint4 w[R][NL];
    __device__ __forceinline__ void issue(const bf16_t* const (&row)[R], int K, int lane) {
        const int K8 = K >> 3;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int kk = lane + 64 * j;
#pragma unroll
            for (int r = 0; r < R; ++r)
                w[r][j] = (kk < K8 && row[r]) ? reinterpret_cast<const uint4*>(row[r])[kk] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __device__ __forceinline__ void dot(const float* x, int K, int lane, float (&acc)[R]) const {
        const int K8 = K >> 3;
        const float4* x4 = reinterpret_cast<const float4*>(x);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int kk = lane + 64 * j;
            if (kk < K8) {
                const float4 xa = x4[2 * kk], xb = x4[2 * kk + 1];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // resident weights (dec_persist_k): an opaque copy keeps the bf16 -> fp32 unpacking INSIDE the frame loop;
                    // hoisted, the unpacked forms double the live set (480 registers) and spill to scratch
                    uint4 t = w[r][j];
                    asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
                    acc[r] += dot8(t, xa, xb);
                }
            }
        }
    }
};

// The same for fp32 weight rows (dec_persist_k<true>: the reference's own inference precision, inference.py:68-71).  A chunk of 8
// weights is two float4; `issue` requests them, `dot` multiplies.  RESIDENT rows are issued once before the frame loop and live in
// registers (AGPRs take what the 256 architectural registers cannot hold: the compiler parks them there and reads them back per
// use); STREAMED rows are re-issued every frame right before the wait for the stage's input and come from the L2 / Infinity Cache.
template <int R, int NL>
struct WRowsF {
    float4 w[R][NL][2];
    __device__ __forceinline__ void issue(const float* const (&row)[R], int K, int lane) {
        const int K8 = K >> 3;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int kk = lane + 64 * j;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = kk < K8 && row[r];
                const float4* src = reinterpret_cast<const float4*>(row[r]) + 2 * kk;
                w[r][j][0] = ok ? src[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                w[r][j][1] = ok ? src[1] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void dot(const float* x, int K, int lane, float (&acc)[R]) const {
        const int K8 = K >> 3;
        const float4* x4 = reinterpret_cast<const float4*>(x);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int kk = lane + 64 * j;
            if (kk < K8) {
                const float4 xa = x4[2 * kk], xb = x4[2 * kk + 1];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 a = w[r][j][0], b = w[r][j][1];
                    acc[r] += (a.x * xa.x + a.y * xa.y + a.z * xa.z + a.w * xa.w) + (b.x * xb.x + b.y * xb.y + b.z * xb.z + b.w * xb.w);
                }
            }
        }
    }
};

template <bool PRECISE = false>
__device__ __forceinline__ void cell_update(const float (&pre)[4], float& c, float& h) {
    float ig, fg, gg, og, cn;
    lstm_cell<!PRECISE>(pre, c, ig, fg, gg, og, cn, h);       // fast: v_exp / v_rcp forms (common.h, 16-bit operand modes); precise: libm
    c = cn;
}
// wave sum by DPP butterflies inside the 16-lane rows + four v_readlane (the ds_bpermute ladder of common.h's wave_sum costs
// ~0.2 us per sum, several sums sit on every stage's critical path); the result is wave-uniform
__device__ __forceinline__ float wsum(float v) {
    auto step = [](float x, auto ctrl) {
        return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v = step(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});    // row_mirror
    const unsigned b = __float_as_uint(v);
    return (__uint_as_float(__builtin_amdgcn_readlane(b, 0)) + __uint_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__uint_as_float(__builtin_amdgcn_readlane(b, 32)) + __uint_as_float(__builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float sfloat(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * 1.4426950408889634f * x) + 1.f);
}

#ifndef FT_DECODE_LIBM
#define FT_DECODE_LIBM 0
#endif
template <int R, int NL, bool F32> using template_rows = typename std::conditional<F32, WRowsF<R, NL>, WRows<R, NL>>::type;

// F32 = false: 16-bit weight images, all of them register-resident (the round-2 kernel).
// F32 = true (round 4): fp32 weights and fp32 FMAs -- the operand precision of the reference's inference.py:68-71; activations by the
//   v_exp_f32 / v_rcp_f32 forms (|err| ~ 1e-7, mel 2.4e-7 from the fp32 CPU restatement over 400 frames; -DFT_DECODE_LIBM=1 selects libm: +10 us per frame).  107 MB per flow do
//   not fit the register file (419 KB per CU against 512 KB of registers less the working set): the five recurrent / large input
//   matrices of the LSTMs stay RESIDENT -- attention W_hh, layer-0 W_ih[:, :H] and W_hh, layer-1 W_hh in registers (256 per lane =
//   the accumulation half of the register file, where the compiler parks them), layer-1 W_ih and layer-0 W_ih[:, H:] in 104 KB of
//   LDS --, the rest -- attention W_ih, the query rows, the two dense layers, the 1x1 conv: 37 KB per wave and frame, most of it
//   replicated per XCD and L2-resident -- is STREAMED, requested right before the wait for the stage's input.
template <bool F32>
__global__ __launch_bounds__(256, 1) void dec_persist_k(const DecP p) {
    const DecodeDev& P = p.d;
    constexpr int H = 1024, A = 640, M = 80, LMAX = 1024;
    __shared__ __attribute__((aligned(16))) float s_prev[M + 16], s_cat2[2][H + A], s_q[A], s_pr[LMAX], s_h0b[2][H],
        s_h1b[2][H], s_u1[H], s_u2[H], s_o[2 * M + 16], s_v[A], s_gw[H + A];
    __shared__ float s_red[8];
    // fp32 mode: layer-1 W_ih and layer-0 W_ih[:, H:] of this workgroup's four units live in LDS for the whole utterance
    // ([wave][gate][H] then [wave][gate][A] floats: 104 KB of dynamic LDS) -- as streamed operands they were 27 MB per frame and flow
    // out of the Infinity Cache on the two LSTM stages' critical paths (measured 4.4 / 5.0 us for those stages against 2.4 / 3.2)
    extern __shared__ __attribute__((aligned(16))) float s_wlds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = blockIdx.x;
    const int u = c * 4 + wave;                                   // this wave's hidden unit / dense row
    float* const s_w1 = s_wlds + (size_t)wave * 4 * H;            // [gate][H]
    float* const s_w0c = s_wlds + (size_t)4 * 4 * H + (size_t)wave * 4 * A;   // [gate][A]
    if constexpr (F32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4* src1 = reinterpret_cast<const float4*>(P.l1_w_ih + ((size_t)g * H + u) * H);
            for (int k4 = lane; k4 < H / 4; k4 += 64) reinterpret_cast<float4*>(s_w1 + g * H)[k4] = src1[k4];
            const float4* src0 = reinterpret_cast<const float4*>(P.l0_w_ih + ((size_t)g * H + u) * (H + A) + H);
            for (int k4 = lane; k4 < A / 4; k4 += 64) reinterpret_cast<float4*>(s_w0c + g * A)[k4] = src0[k4];
        }
    }
    // four rows of K floats in LDS times the activation vector x (same chunking as WRowsF::dot)
    auto dot_lds = [&](const float* wl, const float* x, int K, float (&acc)[4]) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (int kk = lane; kk < (K >> 3); kk += 64) {
            const float4 xa = x4[2 * kk], xb = x4[2 * kk + 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 a = reinterpret_cast<const float4*>(wl + (size_t)r * K)[2 * kk], b = reinterpret_cast<const float4*>(wl + (size_t)r * K)[2 * kk + 1];
                acc[r] += (a.x * xa.x + a.y * xa.y + a.z * xa.z + a.w * xa.w) + (b.x * xb.x + b.y * xb.y + b.z * xb.z + b.w * xb.w);
            }
        }
    };
    const int L = P.L, N = P.N;
    const long t_start = wall_clock64();
    __shared__ int s_slot[2];
    if (tid == 0) {                                               // XCD census: which L2 this workgroup shares, and its rank there
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        s_slot[0] = (int)(xcc & 7u);
        s_slot[1] = (int)__hip_atomic_fetch_add(p.census + (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // recurrent inputs double-buffered by frame parity: frame i reads the vectors frame i - 1 gathered (no roll copy)
    for (int k = tid; k < H; k += 256) { s_cat2[1][k] = 0.f; s_h0b[1][k] = 0.f; s_h1b[1][k] = 0.f; }
    __syncthreads();
    Relay R;
    R.glob = p.gran;
    R.loc = p.gran + (size_t)G_TOTAL * (1 + __builtin_amdgcn_readfirstlane(s_slot[0]));
    R.q = __builtin_amdgcn_readfirstlane(s_slot[1]);
    if (R.q >= 32) {                                              // not 32 workgroups per XCD: not the machine this is for
        if (tid == 0) atomicExch(p.status, 2);
        return;
    }
    float c_att = 0.f, c_0 = 0.f, c_1 = 0.f;                      // cell states of unit u (lane 0 of the wave is the keeper)
    int i = 0, done = 0;
    typedef typename std::conditional<F32, float, bf16_t>::type wt_t;
    const wt_t* rows4[4];
    auto gate_rows = [&](const wt_t* W, int K, int col0 = 0) {    // the four gate rows of unit u (from column col0 on)
#pragma unroll
        for (int g = 0; g < 4; ++g) rows4[g] = W + ((size_t)g * H + u) * K + col0;
    };
    auto wsel = [](const bf16_t* w16, const float* w32) -> const wt_t* {
        if constexpr (F32) return w32; else return w16;
    };
    // ---- 16-bit mode: the flow's weights live in REGISTERS for all N frames: 53.7 MB of bf16 over 256 CUs x 4 waves = 205 KB per CU
    // = ~240 VGPRs per lane (one wave per SIMD owns the whole 512-entry file); nothing is streamed per frame but the hand-offs.
    // fp32 mode: see the kernel's head comment (resident: wa_hh, w0_ih over h_att, w0_hh, w1_hh; the rest streamed).
    template_rows<4, 1, F32> wa_ih;  template_rows<4, 2, F32> wa_hh;  template_rows<5, 2, F32> wq;
    template_rows<4, F32 ? 2 : 4, F32> w0_ih;                     // (fp32 mode: the h_att columns; the ctx columns sit in LDS)
    template_rows<4, 2, F32> w0_hh;  template_rows<4, 2, false> w1_ih;  template_rows<4, 2, F32> w1_hh;   // (w1_ih: 16-bit mode only)
    template_rows<1, 2, F32> wd0, wd1;  template_rows<2, 2, F32> wcv;
    // The small stages (query, scores, context, 1x1 conv) are computed by EVERY XCD for itself -- 8x redundant arithmetic on
    // resident operands -- so their hand-offs never leave the XCD's L2 (~0.5 us instead of ~2.5 us through the fabric).
    // Wave `slot` of the XCD's 128 waves takes query rows / context channels slot + 128 k, text positions slot + 128 k and
    // conv rows slot, slot + 128.
    const int slot = R.q * 4 + wave;
    const wt_t* r5[5];
    const wt_t* r2[2];
    const wt_t* r1a[1];
    const wt_t* r1b[1];
#pragma unroll
    for (int k = 0; k < 5; ++k) r5[k] = wsel(P.w_query16, P.w_query) + (size_t)(slot + 128 * k) * H;
    r2[0] = wsel(P.conv_w16, P.conv_w) + (size_t)slot * H;
    r2[1] = slot + 128 < 2 * M ? wsel(P.conv_w16, P.conv_w) + (size_t)(slot + 128) * H : nullptr;
    r1a[0] = wsel(P.d0_w16, P.d0_w) + (size_t)u * H;
    r1b[0] = wsel(P.d1_w16, P.d1_w) + (size_t)u * H;
    // streamed (fp32 mode) weight requests, each placed right before the wait for its stage's input
    auto issue_att_ih = [&]() { gate_rows(wsel(P.att_w_ih16, P.att_w_ih), M); wa_ih.issue(rows4, M, lane); };
    gate_rows(wsel(P.att_w_hh16, P.att_w_hh), H); wa_hh.issue(rows4, H, lane);
    if constexpr (F32) { gate_rows(P.l0_w_ih, H + A); w0_ih.issue(rows4, H, lane); }
    else { gate_rows(wsel(P.l0_w_ih16, P.l0_w_ih), H + A); w0_ih.issue(rows4, H + A, lane); }
    gate_rows(wsel(P.l0_w_hh16, P.l0_w_hh), H); w0_hh.issue(rows4, H, lane);
    gate_rows(wsel(P.l1_w_hh16, P.l1_w_hh), H); w1_hh.issue(rows4, H, lane);
    if constexpr (!F32) {                                         // 16-bit mode: everything resident
        gate_rows(P.l1_w_ih16, H); w1_ih.issue(rows4, H, lane);
        issue_att_ih();
        wq.issue(r5, H, lane);
        wcv.issue(r2, H, lane);
        wd0.issue(r1a, H, lane);
        wd1.issue(r1b, H, lane);
    }
    float b_att[4], b_0[4], b_1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const size_t r = (size_t)g * H + u;
        // wave-uniform: scalar registers
        b_att[g] = sfloat(P.att_b_ih[r] + P.att_b_hh[r]); b_0[g] = sfloat(P.l0_b_ih[r] + P.l0_b_hh[r]); b_1[g] = sfloat(P.l1_b_ih[r] + P.l1_b_hh[r]);
    }
    const float b_d0 = sfloat(P.d0_b[u]), b_d1 = sfloat(P.d1_b[u]);
    const float b_cv[2] = {sfloat(P.conv_b[slot]), sfloat(slot + 128 < 2 * M ? P.conv_b[slot + 128] : 0.f)};
    // frame-invariant attention operands of this wave: key row / v (text position c + 256 wave), value column (channel qrow),
    // the gate row (workgroup 0)
    // resident: key rows of positions slot, slot + 128 and value columns over l < 256 (texts up to 256 symbols never touch
    // memory for them); longer texts read the rest from the XCD's L2 each frame
    // (fp32 mode keeps half as many: its registers go to the resident LSTM matrices)
    constexpr int KRES = F32 ? 1 : 2, VRES = F32 ? 2 : 4;         // (fp32: texts up to 128 symbols; the rest comes from the L2)
    float k_row[KRES ? KRES : 1][A / 64], v_col[5][VRES ? VRES : 1];
#pragma unroll
    for (int j = 0; j < A / 64; ++j) {
#pragma unroll
        for (int k = 0; k < KRES; ++k) k_row[k][j] = slot + 128 * k < L ? P.K[(size_t)(slot + 128 * k) * A + lane + 64 * j] : 0.f;
    }
    for (int k = tid; k < A; k += 256) s_v[k] = P.v[k];
    for (int k = tid; k < H + A; k += 256) s_gw[k] = (c == 0 && P.gate_w) ? P.gate_w[k] : 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < VRES; ++j) v_col[k][j] = lane + 64 * j < L ? P.V[(size_t)(lane + 64 * j) * A + slot + 128 * k] : 0.f;
    const float gate_b = (c == 0 && P.gate_w) ? P.gate_b[0] : 0.f;
    // activations: the v_exp_f32 / v_rcp_f32 forms in BOTH modes (abs error ~1e-7 = fp32 rounding level; libm's expf / tanhf cost
    // ~100 instructions each on the stage's critical path: 6 per LSTM cell, 10 per score lane -- measured 6 us per frame and flow);
    // -DFT_DECODE_LIBM=1 selects libm in fp32 mode
    constexpr bool LIBM = F32 && FT_DECODE_LIBM;
    auto act_tanh = [](float x) { if constexpr (LIBM) return tanhf(x); else return fast_tanh(x); };
    const bool prof = p.prof != nullptr && c == 0 && tid == 0;
    auto stamp = [&](int k) { if (prof && i < 512) p.prof[(size_t)i * 12 + k] = wall_clock64(); };
    for (;; ++i) {
        const unsigned e0 = (unsigned)i * 16u;                    // epochs of frame i: e0 + 1 .. e0 + 9
        float* const s_cat = s_cat2[i & 1];
        const float* const s_hatt = s_cat2[(i & 1) ^ 1];
        float* const s_h0n = s_h0b[i & 1];
        const float* const s_h0 = s_h0b[(i & 1) ^ 1];
        float* const s_h1n = s_h1b[i & 1];
        const float* const s_h1 = s_h1b[(i & 1) ^ 1];
        stamp(0);
        // ================= S1: inverse coupling of frame i-1 (needs its conv output o), then the attention LSTM of frame i
        if constexpr (F32) issue_att_ih();
        if (i > 0) {
            const float z = tid < M ? P.residual[(size_t)(i - 1) * M + tid] : 0.f;     // requested before the wait
            if (!gather(R, G_O, 2 * M + 1, 2 * M, e0 - 16u + 9u, s_o, p, t_start)) return;
            stamp(1);
            done = s_o[2 * M] != 0.f;
            if (tid < M) {
                const float x = (z - s_o[M + tid]) / expf(s_o[tid]);
                s_prev[tid] = x;
                if (c == 0) P.mel_out[(size_t)(i - 1) * M + tid] = x;
            }
            if (c == 0 && tid == 0) P.n_done_dev[0] = i;
        } else if (tid < M) s_prev[tid] = 0.f;
        __syncthreads();
        if (i >= N || done) break;
        {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            wa_ih.dot(s_prev, M, lane, acc);
            wa_hh.dot(s_hatt, H, lane, acc);
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = wsum(acc[g]) + b_att[g];
            float h;
            cell_update<LIBM>(pre, c_att, h);
            if (lane == 0) publish(p.gran + G_HATT + u, e0 + 1u, h);
        }
        // ================= S2: query rows c, c + 256, c + 512 (waves 0..2)
        stamp(2);
        if constexpr (F32) wq.issue(r5, H, lane);
        if (!gather(R, G_HATT, H, 0, e0 + 1u, s_cat, p, t_start)) return;       // new h_att = first part of [h_att ; ctx]
        stamp(3);
        {
            float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            wq.dot(s_cat, H, lane, acc);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float v = wsum(acc[k]);
                if (lane == 0) publish_local(R.loc + G_Q + slot + 128 * k, e0 + 2u, v);
            }
        }
        // ================= S3a: scores of text positions c, c + 256, ..  (wave w takes position c + 256 w)
        if (!gather(R, G_Q, A, A, e0 + 2u, s_q, p, t_start)) return;
        stamp(4);
#pragma unroll
        for (int k = 0; k < KRES; ++k)
            if (slot + 128 * k < L) {
                float sc = 0.f;
#pragma unroll
                for (int j = 0; j < A / 64; ++j) sc += s_v[lane + 64 * j] * act_tanh(s_q[lane + 64 * j] + k_row[k][j]);
                sc = wsum(sc);
                if (lane == 0) publish_local(R.loc + G_SC + slot + 128 * k, e0 + 3u, sc * P.inv_temp);
            }
        for (int l = slot + 128 * KRES; l < L; l += 128) {         // texts longer than 256 symbols
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < A / 64; ++j) sc += s_v[lane + 64 * j] * act_tanh(s_q[lane + 64 * j] + P.K[(size_t)l * A + lane + 64 * j]);
            sc = wsum(sc);
            if (lane == 0) publish_local(R.loc + G_SC + l, e0 + 3u, sc * P.inv_temp);
        }
        // ================= S3b: softmax over L (every workgroup), context channels c, c + 256, c + 512
        if (!gather(R, G_SC, L, L, e0 + 3u, s_pr, p, t_start)) return;
        stamp(5);
        {
            float m = -INFINITY;
            for (int l = tid; l < L; l += 256) m = fmaxf(m, s_pr[l]);
            m = wave_max(m);
            if (lane == 0) s_red[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
            float sum = 0.f;
            for (int l = tid; l < L; l += 256) { const float e = expf(s_pr[l] - m); s_pr[l] = e; sum += e; }
            sum = wsum(sum);
            if (lane == 0) s_red[4 + wave] = sum;
            __syncthreads();
            sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
            for (int l = tid; l < L; l += 256) {
                const float pl = s_pr[l] / sum;
                s_pr[l] = pl;
                if (c == 0) P.attn_out[(size_t)i * L + l] = pl;
            }
            __syncthreads();
            float pl[VRES ? VRES : 1];
#pragma unroll
            for (int j = 0; j < VRES; ++j) pl[j] = lane + 64 * j < L ? s_pr[lane + 64 * j] : 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                float cx = 0.f;
#pragma unroll
                for (int j = 0; j < VRES; ++j) cx += pl[j] * v_col[k][j];
                for (int l = lane + 64 * VRES; l < L; l += 64) cx += s_pr[l] * P.V[(size_t)l * A + slot + 128 * k];
                cx = wsum(cx);
                if (lane == 0) publish_local(R.loc + G_CTX + slot + 128 * k, e0 + 4u, cx);
            }
        }
        // ================= S4: LSTM layer 0 (input [h_att ; ctx], recurrent h0); workgroup 0 also evaluates the gate
        if (!gather(R, G_CTX, A, A, e0 + 4u, s_cat + H, p, t_start)) return;
        stamp(6);
        float gate_done = 0.f;
        if (c == 0 && P.gate_w) {                                  // flowtron.py:823-826 (uniform branch: all of workgroup 0)
            float g = 0.f;
            for (int k = tid; k < H + A; k += 256) g += s_gw[k] * s_cat[k];
            g = wsum(g);
            if (lane == 0) s_red[wave] = g;
            __syncthreads();
            const float gs = gate_b + s_red[0] + s_red[1] + s_red[2] + s_red[3];
            gate_done = (1.f / (1.f + expf(-gs)) > P.gate_threshold) ? 1.f : 0.f;
        }
        {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (F32) { w0_ih.dot(s_cat, H, lane, acc); dot_lds(s_w0c, s_cat + H, A, acc); }
            else w0_ih.dot(s_cat, H + A, lane, acc);
            w0_hh.dot(s_h0, H, lane, acc);
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = wsum(acc[g]) + b_0[g];
            float h;
            cell_update<LIBM>(pre, c_0, h);
            if (lane == 0) publish(p.gran + G_H0 + u, e0 + 5u, h);
        }
        // ================= S5: LSTM layer 1
        if (!gather(R, G_H0, H, 0, e0 + 5u, s_h0n, p, t_start)) return;
        stamp(7);
        {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (F32) dot_lds(s_w1, s_h0n, H, acc);
            else w1_ih.dot(s_h0n, H, lane, acc);
            w1_hh.dot(s_h1, H, lane, acc);
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = wsum(acc[g]) + b_1[g];
            float h;
            cell_update<LIBM>(pre, c_1, h);
            if (lane == 0) publish(p.gran + G_H1 + u, e0 + 6u, h);
        }
        // ================= S6 / S7: dense + tanh, row u
        if constexpr (F32) wd0.issue(r1a, H, lane);
        if (!gather(R, G_H1, H, 0, e0 + 6u, s_h1n, p, t_start)) return;
        stamp(8);
        {
            float acc[1] = {0.f};
            wd0.dot(s_h1n, H, lane, acc);
            const float v = act_tanh(wsum(acc[0]) + b_d0);
            if (lane == 0) publish(p.gran + G_U1 + u, e0 + 7u, v);
        }
        if constexpr (F32) wd1.issue(r1b, H, lane);
        if (!gather(R, G_U1, H, 0, e0 + 7u, s_u1, p, t_start)) return;
        stamp(9);
        {
            float acc[1] = {0.f};
            wd1.dot(s_u1, H, lane, acc);
            const float v = act_tanh(wsum(acc[0]) + b_d1);
            if (lane == 0) publish(p.gran + G_U2 + u, e0 + 8u, v);
        }
        // ================= S8: 1x1 conv row c (wave 0 of workgroups c < 2M); workgroup 0 appends the stop flag
        if constexpr (F32) wcv.issue(r2, H, lane);
        if (!gather(R, G_U2, H, 0, e0 + 8u, s_u2, p, t_start)) return;
        stamp(10);
        {
            float acc[2] = {0.f, 0.f};
            wcv.dot(s_u2, H, lane, acc);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float v = wsum(acc[k]) + b_cv[k];
                if (lane == 0 && slot + 128 * k < 2 * M) publish_local(R.loc + G_O + slot + 128 * k, e0 + 9u, v);
            }
        }
        if (c == 0 && tid == 0) publish(p.gran + G_O + 2 * M, e0 + 9u, gate_done);

    }
}

struct Layout {
    size_t off_dev, off_state, n_state, off_ctl, total;
    size_t h_att, c_att, h0, c0, h1, c1, q, ctx, u1, u2, prev, cumm, prev_attn, keyin, Kdyn, escore, obuf;
    size_t hx, cx;             // state of the decoder layers beyond the second: [n_layers - 2][2][H], [n_layers - 2][H]
};

Layout make_layout(int H, int A, int M, int L, int E, int n_layers = 2) {
    Layout l{};
    auto up = [](size_t v) { return (v + 63) & ~size_t(63); };
    l.off_dev = 0;
    l.off_state = up(sizeof(DecodeDev));
    size_t f = 0;
    auto take = [&](size_t n) { size_t o = f; f += (n + 15) & ~size_t(15); return o; };
    l.h_att = take(2 * H); l.c_att = take(H); l.h0 = take(2 * H); l.c0 = take(H); l.h1 = take(2 * H); l.c1 = take(H);
    l.q = take(A); l.ctx = take(A); l.u1 = take(H); l.u2 = take(H); l.prev = take(M);
    l.cumm = take(L); l.prev_attn = take(L); l.keyin = take((size_t)L * E); l.Kdyn = take((size_t)L * A);
    l.escore = take(L); l.obuf = take(2 * (size_t)M);
    const size_t nx = n_layers > 2 ? (size_t)(n_layers - 2) : 0;
    l.hx = take(2 * (size_t)H * nx + 16); l.cx = take((size_t)H * nx + 16);
    l.n_state = f;
    l.off_ctl = l.off_state + f * sizeof(float);
    l.total = l.off_ctl + 64;
    return l;
}

constexpr int GRAPH_FRAMES = 8;
std::mutex g_graph_mu;
std::unordered_map<uint64_t, hipGraphExec_t> g_graph_cache;
std::deque<uint64_t> g_graph_order;                  // insertion order: the oldest graph is evicted when the cache is full
constexpr size_t GRAPH_CACHE_MAX = 64;

__global__ void f32_to_bf16_k(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2; i < n; i += (size_t)gridDim.x * blockDim.x * 2) {
        if (i + 1 < n) *reinterpret_cast<unsigned int*>(dst + i) = pack_bf16x2(src[i], src[i + 1]);
        else dst[i] = f2bf(src[i]);
    }
}

// decoder LSTM stack of any depth on the staged chain: layer 0 and 1 are stages 4 / 5 of the parameter block; a layer k >= 2 is stage 5
// again with the block's layer-1 slots pointing at ITS weights and state (every stage kernel takes the block by value), and the
// dense stage reads the last layer's output through the block's h1 slot
struct Depth {
    int n_layers = 2;
    const float* const* extra = nullptr;       // 4 device pointers per layer >= 2
    float *hx = nullptr, *cx = nullptr;
};

int enqueue_frame(const DecodeDev& dP, int H, int A, int L, int M, bool cumm, hipStream_t st, const Depth& dep = Depth()) {
    const dim3 b256(256), b1024(1024);
    const bool w16 = dP.att_w_hh16 != nullptr;
    if (w16) hipLaunchKernelGGL(dec_lstm16_k<0>, dim3(H), b256, 0, st, dP);
    else hipLaunchKernelGGL(dec_lstm_k<0>, dim3(cdiv(H, 4)), b256, 0, st, dP);
    hipLaunchKernelGGL(dec_gemv_k<0>, dim3(cdiv(A, 4)), b256, 0, st, dP);
    if (cumm) {
        hipLaunchKernelGGL(dec_cond_k, dim3(L), b256, 0, st, dP);
        hipLaunchKernelGGL(dec_key_k, dim3(L, cdiv(A, 16)), b256, 0, st, dP);
    }
    hipLaunchKernelGGL(dec_score_k, dim3(cdiv(L, 4)), b256, 0, st, dP);
    hipLaunchKernelGGL(dec_ctx_k, dim3(cdiv(A, 64)), b256, sizeof(float) * (L + 4 + 256), st, dP);
    if (w16) hipLaunchKernelGGL(dec_lstm16_k<1>, dim3(H), b256, 0, st, dP);
    else hipLaunchKernelGGL(dec_lstm_k<1>, dim3(cdiv(H, 4)), b256, 0, st, dP);
    DecodeDev top = dP;                         // the block the dense stage sees: h1 = output of the LAST layer
    if (dep.n_layers == 1) {
        top.h1 = dP.h0;
    } else {
        if (w16) hipLaunchKernelGGL(dec_lstm16_k<2>, dim3(H), b256, 0, st, dP);
        else hipLaunchKernelGGL(dec_lstm_k<2>, dim3(cdiv(H, 4)), b256, 0, st, dP);
        const float* below = dP.h1;
        for (int k = 2; k < dep.n_layers; ++k) {
            DecodeDev lk = dP;
            const float* const* w = dep.extra + 4 * (k - 2);
            lk.l1_w_ih = w[0]; lk.l1_w_hh = w[1]; lk.l1_b_ih = w[2]; lk.l1_b_hh = w[3];
            lk.l1_w_ih16 = nullptr; lk.l1_w_hh16 = nullptr;
            lk.h0 = const_cast<float*>(below);
            lk.h1 = dep.hx + (size_t)(k - 2) * 2 * H; lk.c1 = dep.cx + (size_t)(k - 2) * H;
            hipLaunchKernelGGL(dec_lstm_k<2>, dim3(cdiv(H, 4)), b256, 0, st, lk);
            below = lk.h1;
        }
        top.h1 = const_cast<float*>(below);
    }
    hipLaunchKernelGGL(dec_gemv_k<1>, dim3(cdiv(H, 4)), b256, 0, st, top);
    hipLaunchKernelGGL(dec_gemv_k<2>, dim3(cdiv(H, 4)), b256, 0, st, dP);
    hipLaunchKernelGGL(dec_conv_k, dim3(cdiv(2 * M, 4)), b256, 0, st, dP);
    hipLaunchKernelGGL(dec_fin_k, dim3(1), b256, 0, st, dP);
    return 0;
}

}  // namespace

extern "C" size_t ft_decode_workspace_bytes(int L, int H, int A, int M, int E, int n_layers) {
    return make_layout(H, A, M, L, E, n_layers > 0 ? n_layers : 2).total;
}

namespace {
// element counts of the ten matrices that get a bf16 image, in DecodeDev order
void wimg_counts(int H, int A, int M, size_t (&n)[10]) {
    const size_t H4 = 4 * (size_t)H;
    n[0] = H4 * M; n[1] = H4 * H; n[2] = (size_t)A * H; n[3] = H4 * (H + A); n[4] = H4 * H; n[5] = H4 * H; n[6] = H4 * H;
    n[7] = (size_t)H * H; n[8] = (size_t)H * H; n[9] = 2 * (size_t)M * H;
}
}  // namespace

// the producers' copy + one copy per XCD + the census counters
extern "C" size_t ft_decode_persist_gran_bytes(void) { return (size_t)G_TOTAL * 8 * 9 + 64; }
static long* g_decode_prof = nullptr;
// debug hook: device buffer [512][12] int64 that subsequent persistent decode launches fill with per-frame stage stamps
// (100 MHz wall clock) of workgroup 0; NULL switches it off
extern "C" int ft_decode_debug_prof(void* dev_buf) { g_decode_prof = reinterpret_cast<long*>(dev_buf); return FT_OK; }

extern "C" size_t ft_decode_wimg_bytes(int H, int A, int M) {
    size_t n[10], tot = 0;
    wimg_counts(H, A, M, n);
    for (size_t v : n) tot += (v * 2 + 255) & ~size_t(255);
    return tot;
}

extern "C" int ft_decode_flow(const ft_decode_args* a, void* stream) {
    FT_CHECK_ARG(a != nullptr);
    FT_CHECK_ARG(a->att_w_ih && a->att_w_hh && a->att_b_ih && a->att_b_hh && a->w_query && a->v && a->K && a->V);
    const int n_layers = a->n_layers > 0 ? a->n_layers : 2;
    FT_CHECK_ARG(a->l0_w_ih && a->l0_w_hh && a->l0_b_ih && a->l0_b_hh);
    FT_CHECK_ARG(n_layers == 1 || (a->l1_w_ih && a->l1_w_hh && a->l1_b_ih && a->l1_b_hh));
    FT_CHECK_ARG(n_layers <= 2 || a->extra_layers);
    for (int k = 0; k < 4 * (n_layers - 2); ++k) FT_CHECK_ARG(a->extra_layers[k] != nullptr);
    FT_CHECK_ARG(a->d0_w && a->d0_b && a->d1_w && a->d1_b && a->conv_w && a->conv_b);
    FT_CHECK_ARG((a->gate_w == nullptr) == (a->gate_b == nullptr));
    FT_CHECK_ARG(a->residual && a->mel_out && a->attn_out && a->n_done_dev && a->work);
    FT_CHECK_ARG(a->N >= 0 && a->L >= 1 && a->H >= 1 && a->A >= 1 && a->M >= 1 && a->temperature > 0.f);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(a->work) % 64 == 0);
    const bool cumm = a->cond_w1 != nullptr;
    FT_CHECK_ARG(!cumm || (a->cond_b1 && a->cond_w2 && a->cond_b2 && a->w_key && a->enc && a->E >= 1));
    const Layout lay = make_layout(a->H, a->A, a->M, a->L, cumm ? a->E : 1, n_layers);
    FT_CHECK_ARG(a->work_bytes >= lay.total);
    if (sizeof(float) * ((size_t)a->L + 4 + 256) > 160 * 1024)
        return ft_fail(FT_EUNSUPPORTED, "ft_decode_flow: L=%d exceeds the LDS probability tile", a->L);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(a->work);
    float* fs = reinterpret_cast<float*>(base + lay.off_state);

    DecodeDev h{};
    h.att_w_ih = a->att_w_ih; h.att_w_hh = a->att_w_hh; h.att_b_ih = a->att_b_ih; h.att_b_hh = a->att_b_hh;
    h.w_query = a->w_query; h.v = a->v; h.K = a->K; h.V = a->V;
    h.l0_w_ih = a->l0_w_ih; h.l0_w_hh = a->l0_w_hh; h.l0_b_ih = a->l0_b_ih; h.l0_b_hh = a->l0_b_hh;
    h.l1_w_ih = a->l1_w_ih; h.l1_w_hh = a->l1_w_hh; h.l1_b_ih = a->l1_b_ih; h.l1_b_hh = a->l1_b_hh;
    h.d0_w = a->d0_w; h.d0_b = a->d0_b; h.d1_w = a->d1_w; h.d1_b = a->d1_b; h.conv_w = a->conv_w; h.conv_b = a->conv_b;
    h.gate_w = a->gate_w; h.gate_b = a->gate_b;
    h.residual = a->residual; h.mel_out = a->mel_out; h.attn_out = a->attn_out; h.n_done_dev = a->n_done_dev;
    h.h_att = fs + lay.h_att; h.c_att = fs + lay.c_att; h.h0 = fs + lay.h0; h.c0 = fs + lay.c0; h.h1 = fs + lay.h1; h.c1 = fs + lay.c1;
    h.q = fs + lay.q; h.ctx = fs + lay.ctx; h.u1 = fs + lay.u1; h.u2 = fs + lay.u2; h.prev = fs + lay.prev;
    h.ctl = reinterpret_cast<int*>(base + lay.off_ctl);
    h.escore = fs + lay.escore; h.obuf = fs + lay.obuf;
    h.prior = a->prior; h.forced = a->forced;
    if (cumm) {
        h.cond_w1 = a->cond_w1; h.cond_b1 = a->cond_b1; h.cond_w2 = a->cond_w2; h.cond_b2 = a->cond_b2; h.w_key = a->w_key; h.enc = a->enc;
        h.cumm = fs + lay.cumm; h.prev_attn = fs + lay.prev_attn; h.keyin = fs + lay.keyin; h.Kdyn = fs + lay.Kdyn;
    }
    h.E = a->E;
    h.N = a->N; h.L = a->L; h.H = a->H; h.A = a->A; h.M = a->M;
    if (a->wimg) {                               // bf16 weight images: rounded once per call (54 MB of writes vs N x 107 MB of reads)
        FT_CHECK_ARG(a->wimg_bytes >= ft_decode_wimg_bytes(a->H, a->A, a->M) && reinterpret_cast<uintptr_t>(a->wimg) % 256 == 0);
        FT_CHECK_ARG(a->H % 8 == 0 && a->A % 8 == 0 && a->M % 8 == 0);
        size_t n[10];
        wimg_counts(a->H, a->A, a->M, n);
        // (depth 1: the layer-1 image slots hold copies of layer 0's recurrent matrix -- never read)
        const float* src[10] = {a->att_w_ih, a->att_w_hh, a->w_query, a->l0_w_ih, a->l0_w_hh, n_layers == 1 ? a->l0_w_hh : a->l1_w_ih,
                                n_layers == 1 ? a->l0_w_hh : a->l1_w_hh, a->d0_w, a->d1_w, a->conv_w};
        const bf16_t** dstp[10] = {&h.att_w_ih16, &h.att_w_hh16, &h.w_query16, &h.l0_w_ih16, &h.l0_w_hh16, &h.l1_w_ih16, &h.l1_w_hh16,
                                   &h.d0_w16, &h.d1_w16, &h.conv_w16};
        char* wp = reinterpret_cast<char*>(a->wimg);
        for (int k = 0; k < 10; ++k) {
            FT_CHECK_ARG(reinterpret_cast<uintptr_t>(src[k]) % 16 == 0);
            bf16_t* d = reinterpret_cast<bf16_t*>(wp);
            hipLaunchKernelGGL(f32_to_bf16_k, dim3(1024), dim3(256), 0, st, src[k], d, n[k]);
            *dstp[k] = d;
            wp += (n[k] * 2 + 255) & ~size_t(255);
        }
    }
    h.inv_temp = 1.0f / a->temperature; h.gate_threshold = a->gate_threshold;

    // state (h, c, prev, frame counter, stop flag) = 0; parameter block = h.  hipMemcpyAsync from
    // pageable host memory stages the bytes before returning, so `h` may live on this stack frame.
    FT_CHECK_HIP(hipMemsetAsync(base + lay.off_state, 0, lay.total - lay.off_state, st));
    FT_CHECK_HIP(hipMemsetAsync(a->n_done_dev, 0, sizeof(int), st));
    const DecodeDev& dP = h;       // passed to every stage kernel by value (kernarg segment)
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_ctx_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    // persistent path: 16-bit images or (round 4) the fp32 originals, the default model geometry, plain attention, every workgroup
    // resident on its own CU
    Depth dep;
    dep.n_layers = n_layers; dep.extra = a->extra_layers; dep.hx = fs + lay.hx; dep.cx = fs + lay.cx;
    if (a->persist_status && n_layers == 2 && !cumm && !a->prior && !a->forced && a->H == 1024 && a->A == 640 && a->M == 80 && a->L <= 1024) {
        static int cus = -1;
        if (cus < 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
        }
        if (cus >= 256) {
            FT_CHECK_ARG(a->persist_gran && reinterpret_cast<uintptr_t>(a->persist_gran) % 16 == 0);
            FT_CHECK_HIP(hipMemsetAsync(a->persist_gran, 0, ft_decode_persist_gran_bytes(), st));   // tags = 0 (epochs start at 1)
            unsigned long long* gr = reinterpret_cast<unsigned long long*>(a->persist_gran);
            DecP dp{h, gr, reinterpret_cast<unsigned*>(gr + (size_t)G_TOTAL * 9), a->persist_status, 100000000L / 2, g_decode_prof};
            if (a->wimg) hipLaunchKernelGGL(dec_persist_k<false>, dim3(256), dim3(256), 0, st, dp);
            else {
                const int lds32 = (int)(sizeof(float) * 4 * 4 * (1024 + 640));           // layer-1 W_ih + layer-0 W_ih[:, H:] rows
                FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds32));
                hipLaunchKernelGGL(dec_persist_k<true>, dim3(256), dim3(256), lds32, st, dp);
            }
            FT_CHECK_LAUNCH();
            return FT_OK;
        }
    }
    if (!a->use_graph) {
        for (int i = 0; i < a->N; ++i) enqueue_frame(dP, a->H, a->A, a->L, a->M, cumm, st, dep);
        FT_CHECK_LAUNCH();
        return FT_OK;
    }
    // hipGraph path: GRAPH_FRAMES frames per graph; kernels past frame N or past the stop flag are no-ops.
    // the graph bakes the by-value parameter block: key = every byte of it (value-initialised, so padding is zero).
    // Callers that keep their buffers persistent (model.AR_Step.infer does) replay the same graph for every utterance.
    uint64_t key = 1469598103934665603ull;
    {
        const unsigned char* c = reinterpret_cast<const unsigned char*>(&h);
        for (size_t k = 0; k < sizeof(DecodeDev); ++k) { key ^= c[k]; key *= 1099511628211ull; }
        key ^= (uint64_t)n_layers; key *= 1099511628211ull;
        for (int k = 0; k < 4 * (n_layers - 2); ++k) { key ^= reinterpret_cast<uintptr_t>(a->extra_layers[k]); key *= 1099511628211ull; }
    }
    hipGraphExec_t exec = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_graph_mu);
        auto it = g_graph_cache.find(key);
        if (it != g_graph_cache.end()) exec = it->second;
        if (!exec) {
            if (g_graph_cache.size() >= GRAPH_CACHE_MAX) {         // long-running servers with varied (N, L, threshold, buffers): evict
                const uint64_t old = g_graph_order.front();
                g_graph_order.pop_front();
                auto o = g_graph_cache.find(old);
                if (o != g_graph_cache.end()) {
                    FT_CHECK_HIP(hipStreamSynchronize(st));           // the evicted graph may still be running on this stream
                    hipGraphExecDestroy(o->second);
                    g_graph_cache.erase(o);
                }
            }
            hipStream_t cs;
            FT_CHECK_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                for (int f = 0; f < GRAPH_FRAMES; ++f) enqueue_frame(dP, a->H, a->A, a->L, a->M, cumm, cs, dep);
                e = hipStreamEndCapture(cs, &graph);
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph) hipGraphDestroy(graph);
            hipStreamDestroy(cs);
            if (e != hipSuccess) return ft_fail(FT_EHIP, "ft_decode_flow: graph capture failed: %s", hipGetErrorString(e));
            g_graph_cache[key] = exec;
            g_graph_order.push_back(key);
        }
    }
    for (int i = 0; i < a->N; i += GRAPH_FRAMES) FT_CHECK_HIP(hipGraphLaunch(exec, st));
    return FT_OK;
}
