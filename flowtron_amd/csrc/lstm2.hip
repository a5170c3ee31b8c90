// Two stacked LSTM layers as ONE launch chain (the decoder `lstm` of AR_Step: nn.LSTM(1664 -> 1024, num_layers = 2),
// reference flowtron.py:654, :760-765).
//
// A recurrence step is a ~5 us launch that is bound by fixed latency (dispatch, one memory-side round trip for the
// fragment images, LDS reduce, epilogue), not by work: the chip is nearly idle while it runs.  Layer 1 at time
// t-1 and layer 0 at time t are independent once h0[t-1] exists, so they are issued as TWO WORKGROUP GROUPS OF THE SAME
// LAUNCH (a software wavefront): T+1 launches drive both layers instead of 2T, and the per-step input projection of
// layer 1 rides along as a second fragment stream -- its A operand is exactly the bf16 fragment image of h0 that layer 0
// wrote one launch earlier, so the batched gx1 GEMM disappears from the forward pass.
//
//   forward  launch s : group 0 = layer 0 step s          gates = gx0[s] + h0[s-1] W_hh0^T
//                       group 1 = layer 1 step s-1        gates = b1 + h0[s-1] W_ih1^T + h1[s-2] W_hh1^T
//   backward launch s : group 0 = layer 1 step s          dh1 = dy1[s] + dgates1[s+1] W_hh1
//   (s = T-1 .. -1)     group 1 = layer 0 step s+1        dh0 = dgates1[s+1] W_ih1 + dgates0[s+2] W_hh0
//
// bf16 fragment-order operands, fp32 accumulate / state / saved tensors (same conventions as lstm.hip, FT_BF16 path).
// Requires H % 128 == 0 and B <= 64; callers fall back to two ft_lstm_seq_* sequences otherwise.
#include <stdlib.h>

#include "common.h"

namespace {

template <int MT, int G, typename Hook>
__device__ __forceinline__ void skinny_bf16(const bf16x8* __restrict__ afrag, const bf16x8* __restrict__ wfrag,
                                            int nchunk, int c0, int cs, int lane, f32x4 (&acc)[MT], Hook&& after_last_loads,
                                            int a_mt) {
    bf16x8 w[G], a[G][MT];
    auto load_group = [&](int cb) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const size_t c = (size_t)(cb + i * cs);
            w[i] = wfrag[c * 64 + lane];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[i][m] = afrag[(c * a_mt + m) * 64 + lane];
        }
    };
    auto mfma_group = [&]() {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma16(a[i][m], w[i], acc[m]);
    };
    load_group(c0);
    for (int cb = c0 + cs * G; cb < nchunk; cb += cs * G) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_group();
        load_group(cb);
    }
    after_last_loads();                      // younger loads: never delay a fragment wait (vmcnt retires in order)
    __builtin_amdgcn_sched_barrier(0);
    mfma_group();
}

// Two K-segments (two activation images, one K-concatenated weight image), one group of G chunks per wave and segment.
// The weight fragments are the cold stream (every workgroup owns a private slice that comes from the memory side each
// launch); the activation images are 64 KiB that all workgroups read, i.e. hot in L2.  So BOTH segments' weight fragments
// are requested up front, and only the second segment's activation fragments -- a short L2 round trip -- are fetched
// after the first segment's MFMAs, into the same registers (two back-to-back skinny_bf16 calls pay the long round trip
// twice; holding both segments' activations as well does not fit the register budget of two workgroups per CU).
template <int MT, int G, typename Hook>
__device__ __forceinline__ void skinny_bf16_dualw(const bf16x8* __restrict__ afrag0, const bf16x8* __restrict__ afrag1,
                                                  const bf16x8* __restrict__ wfrag0, const bf16x8* __restrict__ wfrag1,
                                                  int c0, int cs, int lane, f32x4 (&acc)[MT], Hook&& after_last_loads, int a_mt) {
    bf16x8 w0[G], w1[G], a[G][MT];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const size_t c = (size_t)(c0 + i * cs);
        w0[i] = wfrag0[c * 64 + lane];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[i][m] = afrag0[(c * a_mt + m) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < G; ++i) w1[i] = wfrag1[(size_t)(c0 + i * cs) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma16(a[i][m], w0[i], acc[m]);
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) a[i][m] = afrag1[((size_t)(c0 + i * cs) * a_mt + m) * 64 + lane];
    after_last_loads();                      // younger loads: never delay a fragment wait (vmcnt retires in order)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma16(a[i][m], w1[i], acc[m]);
}

__device__ __forceinline__ size_t frag_index(int b, int k, int MT) {
    const int c = k >> 5, kg = (k >> 3) & 3, e = k & 7, m = b >> 4, li = b & 15;
    return (((size_t)c * MT + m) * 64 + kg * 16 + li) * 8 + e;
}

// forward weight image of ONE layer over a K-concatenation [Wa | Wb] (Wa may be null: K = H only):
//   out[jb][c][lane(kg,li)][e] = W[(li>>2)*H + jb*4 + (li&3)][c*32 + kg*8 + e],  W[r][k] = k < Ka ? Wa[r][k] : Wb[r][k-Ka]
__global__ void make_wfrag_fwd_cat(const float* __restrict__ wa, int Ka, const float* __restrict__ wb, int Kb,
                                   unsigned short* __restrict__ out, int H) {
    const int K = Ka + Kb, nchunk = K >> 5;
    const size_t total = (size_t)4 * H * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t rest = i >> 9;
        const int c = (int)(rest % nchunk), jb = (int)(rest / nchunk);
        const int li = lane & 15, kg = lane >> 4;
        const size_t row = (size_t)(li >> 2) * H + jb * 4 + (li & 3);
        const int k = c * 32 + kg * 8 + e;
        out[i] = f2op16(k < Ka ? wa[row * Ka + k] : wb[row * Kb + (k - Ka)]);
    }
}
// backward image of W [4H][Kc]: out[jt][c][lane][e] = W[c*32 + kg*8 + e][jt*16 + li]   (jt over Kc/16 column tiles)
__global__ void make_wfrag_bwd_t(const float* __restrict__ w, unsigned short* __restrict__ out, int H, int Kc) {
    const size_t total = (size_t)4 * H * Kc;
    const int nchunk = (4 * H) >> 5;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t rest = i >> 9;
        const int c = (int)(rest % nchunk), jt = (int)(rest / nchunk);
        const int li = lane & 15, kg = lane >> 4;
        const size_t r = (size_t)c * 32 + kg * 8 + e;
        out[i] = f2op16(w[r * Kc + jt * 16 + li]);
    }
}

struct L2FwdP {
    const float* gx0; const float* bias1; const int* lens;
    float *c0, *y0, *gates0, *cell0;
    float *c1, *y1, *gates1, *cell1;
    const unsigned short *w0frag, *w1frag;      // [H/4][H/32][64][8], [H/4][2H/32][64][8]
    unsigned short *h0frag[2], *h1frag[2];      // [H/32][MT][64][8] ping-pong
    int s, T, B, H;
};

// NW waves per workgroup split the k-chunks (4: G chunks per wave and segment; 8: half as many, so that the layer-1 group
// can hold BOTH segments' fragments in flight inside the 128-VGPR budget of two 512-thread workgroups per CU)
template <int MT, int G, int NW>
__device__ __forceinline__ void lstm2_fwd_body(const L2FwdP& p) {
    __shared__ float red[NW][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B, nblk = H >> 2;
    const bool L1 = (int)blockIdx.x >= nblk;
    const int blk = L1 ? blockIdx.x - nblk : blockIdx.x;
    const int s = L1 ? p.s - 1 : p.s;                      // this layer's time step
    if (s < 0 || s >= p.T) return;                          // uniform per workgroup
    const int u0 = blk * 4;
    const int nchunk = H >> 5;

    constexpr int NROLE = MT * 64;
    const bool pf_role = (MT <= 2) && tid >= NROLE;
    const int rr = pf_role ? tid - NROLE : tid;
    const int eb = rr >> 2, ul = rr & 3, eu = u0 + ul;
    const bool ev = !pf_role && tid < NROLE && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    float* cst = L1 ? p.c1 : p.c0;
    int len;
    float gxv[4], c_old;
    auto issue_epilogue_loads = [&]() {
        len = p.lens[ebc];
        if (L1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gxv[g] = p.bias1[(size_t)g * H + eu];
        } else {
            const int t_ld = (pf_role && s + 1 < p.T) ? s + 1 : s;          // role-less waves warm L2 with the next gx0 row
            const float* gp = p.gx0 + ((size_t)t_ld * B + ebc) * 4 * H + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) gxv[g] = gp[(size_t)g * H];
        }
        c_old = cst[(size_t)ebc * H + eu];
    };

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!L1) {
        skinny_bf16<MT, G>(reinterpret_cast<const bf16x8*>(p.h0frag[s & 1]),
                           reinterpret_cast<const bf16x8*>(p.w0frag) + (size_t)blk * nchunk * 64, nchunk, wave, NW, lane, acc,
                           issue_epilogue_loads, MT);
    } else {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.w1frag) + (size_t)blk * 2 * nchunk * 64;
        // input segment: h0 of THIS time step = what layer 0 wrote one launch ago = the buffer layer 0 reads in this launch;
        // recurrent segment: h1 of the previous step
        const bf16x8* a0 = reinterpret_cast<const bf16x8*>(p.h0frag[(s + 1) & 1]);
        const bf16x8* a1 = reinterpret_cast<const bf16x8*>(p.h1frag[s & 1]);
        if (nchunk == NW * G) {             // one group per wave and segment (H = 1024 with G = 8): both weight groups up front
            skinny_bf16_dualw<MT, G>(a0, a1, wf, wf + (size_t)nchunk * 64, wave, NW, lane, acc, issue_epilogue_loads, MT);
        } else {
            skinny_bf16<MT, G>(a0, wf, nchunk, wave, NW, lane, acc, []() {}, MT);
            skinny_bf16<MT, G>(a1, wf + (size_t)nchunk * 64, nchunk, wave, NW, lane, acc, issue_epilogue_loads, MT);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + kg * 4 + r][li] = acc[m][r];
    __syncthreads();
    asm volatile("" ::"v"(gxv[0]), "v"(gxv[1]), "v"(gxv[2]), "v"(gxv[3]), "v"(c_old));
    if (!ev) return;

    const bool active = s < len;
    float* y = L1 ? p.y1 : p.y0;
    unsigned short* hnext = L1 ? p.h1frag[(s + 1) & 1] : p.h0frag[(s + 1) & 1];
    const unsigned short* hprev = L1 ? p.h1frag[s & 1] : p.h0frag[s & 1];
    const size_t row = (size_t)s * B + eb;
    if (!active) {                                           // finished sample: zero pad row, frozen state
        y[row * H + eu] = 0.f;
        hnext[frag_index(eb, eu, MT)] = hprev[frag_index(eb, eu, MT)];
        return;
    }
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = g * 4 + ul;
        float sum = gxv[g];
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[w][eb][n];
        pre[g] = sum;
    }
    float ig, fg, gg, og, c_new, h_new;
    lstm_cell<true>(pre, c_old, ig, fg, gg, og, c_new, h_new);
    cst[(size_t)eb * H + eu] = c_new;
    hnext[frag_index(eb, eu, MT)] = f2op16(h_new);
    y[row * H + eu] = h_new;
    float* gp = (L1 ? p.gates1 : p.gates0) + row * 4 * H + eu;
    gp[0] = ig; gp[(size_t)H] = fg; gp[(size_t)2 * H] = gg; gp[(size_t)3 * H] = og;
    (L1 ? p.cell1 : p.cell0)[row * H + eu] = c_new;
}

// Both layers in ONE workgroup per 4 hidden units (grid H/4, one workgroup per CU at H = 1024).  The step kernels are bound
// by the bytes each CU pulls through its vector-memory path (~40 GB/s per CU whatever the source: the chunk-skew experiment
// priced a launch at ~2.5 us + 0.33 us per MB per... see DESIGN.md), and in the two-group layout every CU hosts one workgroup
// of each group: 32 + 64 KiB of weight fragments plus THREE 64 KiB activation images (h0 twice, h1 once).  Here a wave
// loads its h0 fragments once and feeds them to both W_hh0 (layer 0, step s) and W_ih1 (layer 1, step s-1): 224 KiB per CU
// instead of 288.  Same per-wave accumulation order as the two-group kernel, hence bit-identical results.
// Threads [0, MT*64) finish layer 0, [MT*64, 2*MT*64) layer 1 (MT <= 2).  One group of G = H/128 chunks per wave.
template <int MT, int G>
__global__ __launch_bounds__(256) void lstm2_fwd_both(L2FwdP p) {
    __shared__ float red[2][4][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B;
    const int blk = blockIdx.x, u0 = blk * 4, nchunk = H >> 5;
    const int s0 = p.s, s1 = p.s - 1;                        // layer 0 / layer 1 time step of this launch
    const bool on0 = s0 < p.T, on1 = s1 >= 0;                // uniform

    constexpr int NROLE = MT * 64;
    const bool L1 = tid >= NROLE;
    const int rr = L1 ? tid - NROLE : tid;
    const int eb = rr >> 2, ul = rr & 3, eu = u0 + ul;
    const int s = L1 ? s1 : s0;
    const bool ev = tid < 2 * NROLE && eb < B && (L1 ? on1 : on0);
    const int ebc = eb < B ? eb : B - 1;
    float* cst = L1 ? p.c1 : p.c0;
    int len;
    float gxv[4], c_old;
    auto issue_epilogue_loads = [&]() {
        len = p.lens[ebc];
        if (L1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gxv[g] = p.bias1[(size_t)g * H + eu];
        } else {
            const int t_ld = on0 ? s0 : p.T - 1;
            const float* gp = p.gx0 + ((size_t)t_ld * B + ebc) * 4 * H + eu;
#pragma unroll
            for (int g = 0; g < 4; ++g) gxv[g] = gp[(size_t)g * H];
        }
        c_old = cst[(size_t)ebc * H + eu];
    };

    f32x4 acc0[MT], acc1[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc0[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    {
        const bf16x8* a0 = reinterpret_cast<const bf16x8*>(p.h0frag[s0 & 1]);          // h0[s0-1]: layer 0 recurrent = layer 1 input
        const bf16x8* a1 = reinterpret_cast<const bf16x8*>(p.h1frag[(s1 + 2) & 1]);    // h1[s1-1]  ((s1+2)&1 == s1&1, >= 0)
        const bf16x8* wf0 = reinterpret_cast<const bf16x8*>(p.w0frag) + (size_t)blk * nchunk * 64;
        const bf16x8* wfi = reinterpret_cast<const bf16x8*>(p.w1frag) + (size_t)blk * 2 * nchunk * 64;
        const bf16x8* wf1 = wfi + (size_t)nchunk * 64;
        // one workgroup per CU, one wave per SIMD: the register file is this wave's alone, so EVERY fragment of the step --
        // three weight groups and both activation images -- is requested before the first MFMA (one memory round trip)
        bf16x8 w0[G], wi[G], w1[G], a[G][MT], ah[G][MT];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const size_t c = (size_t)(wave + i * 4);
            w0[i] = wf0[c * 64 + lane];
            wi[i] = wfi[c * 64 + lane];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[i][m] = a0[(c * MT + m) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const size_t c = (size_t)(wave + i * 4);
            w1[i] = wf1[c * 64 + lane];
#pragma unroll
            for (int m = 0; m < MT; ++m) ah[i][m] = a1[(c * MT + m) * 64 + lane];
        }
        issue_epilogue_loads();                  // younger loads: never delay a fragment wait (vmcnt retires in order)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc0[m] = mfma16(a[i][m], w0[i], acc0[m]);
                acc1[m] = mfma16(a[i][m], wi[i], acc1[m]);
            }
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc1[m] = mfma16(ah[i][m], w1[i], acc1[m]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[0][wave][m * 16 + kg * 4 + r][li] = acc0[m][r];
            red[1][wave][m * 16 + kg * 4 + r][li] = acc1[m][r];
        }
    __syncthreads();
    asm volatile("" ::"v"(gxv[0]), "v"(gxv[1]), "v"(gxv[2]), "v"(gxv[3]), "v"(c_old));
    if (!ev) return;

    const bool active = s < len;
    float* y = L1 ? p.y1 : p.y0;
    unsigned short* hnext = L1 ? p.h1frag[(s + 1) & 1] : p.h0frag[(s + 1) & 1];
    const unsigned short* hprev = L1 ? p.h1frag[s & 1] : p.h0frag[s & 1];
    const size_t row = (size_t)s * B + eb;
    if (!active) {                                           // finished sample: zero pad row, frozen state
        y[row * H + eu] = 0.f;
        hnext[frag_index(eb, eu, MT)] = hprev[frag_index(eb, eu, MT)];
        return;
    }
    const int lay = L1 ? 1 : 0;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = g * 4 + ul;
        float sum = gxv[g];
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[lay][w][eb][n];
        pre[g] = sum;
    }
    float ig, fg, gg, og, c_new, h_new;
    lstm_cell<true>(pre, c_old, ig, fg, gg, og, c_new, h_new);
    cst[(size_t)eb * H + eu] = c_new;
    hnext[frag_index(eb, eu, MT)] = f2op16(h_new);
    y[row * H + eu] = h_new;
    float* gp = (L1 ? p.gates1 : p.gates0) + row * 4 * H + eu;
    gp[0] = ig; gp[(size_t)H] = fg; gp[(size_t)2 * H] = gg; gp[(size_t)3 * H] = og;
    (L1 ? p.cell1 : p.cell0)[row * H + eu] = c_new;
}

struct L2BwdP {
    const float* dy1; const int* lens;
    const float *gates1, *cell1, *gates0, *cell0;
    float *dc1, *dc0, *dgx1, *dgx0;
    const unsigned short *wT1frag, *wT0frag, *wTi1frag;     // W_hh1^T, W_hh0^T, W_ih1^T images [H/16][4H/32][64][8]
    unsigned short *da1frag[2], *da0frag[2];                // dgates images over K = 4H: [4H/32][MT][64][8]
    float* pbuf[2];                                         // skew-2 kernel: dgates1[t] W_ih1 for layer 0 step t, [MT*16][H] fp32 ping-pong
    int s, T, B, H, MT;
};

// grid = (2 * H/16, MT): x < H/16 -> layer 1 step s ; x >= H/16 -> layer 0 step s+1.  One 16-row batch tile per workgroup.
template <int G>
__device__ __forceinline__ void lstm2_bwd_body(const L2BwdP& p) {
    __shared__ float red[16][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B, ntile = H >> 4;
    const bool L0 = (int)blockIdx.x >= ntile;
    const int jt = L0 ? blockIdx.x - ntile : blockIdx.x;
    const int s = L0 ? p.s + 1 : p.s;                       // this layer's time step
    if (s < 0 || s >= p.T) return;
    const int j0 = jt * 16;
    const int m_base = blockIdx.y, b_base = m_base * 16;
    const int nchunk = (4 * H) >> 5;

    const bool pf_role = tid >= 256;
    const int rr = pf_role ? tid - 256 : tid;
    const int ebl = rr >> 4, eb = b_base + ebl, jl = rr & 15, eu = j0 + jl;
    const bool ev = !pf_role && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    const float* gates = L0 ? p.gates0 : p.gates1;
    const float* cell = L0 ? p.cell0 : p.cell1;
    float* dcar = L0 ? p.dc0 : p.dc1;
    int len;
    float ig, fg, gg, og, c_t, c_prev, dyv, dcc;
    auto issue_epilogue_loads = [&]() {
        len = p.lens[ebc];
        const int t_ld = (pf_role && s >= 1) ? s - 1 : s;               // role-less waves: next launch's rows
        int tp = t_ld - 1;
        tp = tp < 0 ? 0 : tp;
        const size_t row = (size_t)t_ld * B + ebc;
        const float* gp = gates + row * 4 * H + eu;
        ig = gp[0]; fg = gp[(size_t)H]; gg = gp[(size_t)2 * H]; og = gp[(size_t)3 * H];
        c_t = cell[row * H + eu];
        c_prev = cell[((size_t)tp * B + ebc) * H + eu];
        dyv = L0 ? 0.f : p.dy1[row * H + eu];                            // layer 0 feeds only layer 1: no external dy
        dcc = dcar[(size_t)ebc * H + eu];
    };

    f32x4 acc[1];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!L0) {
        skinny_bf16<1, G>(reinterpret_cast<const bf16x8*>(p.da1frag[(s + 1) & 1]) + (size_t)m_base * 64,
                          reinterpret_cast<const bf16x8*>(p.wT1frag) + (size_t)jt * nchunk * 64, nchunk, wave, 16, lane, acc,
                          issue_epilogue_loads, p.MT);
    } else {
        // dh0[s] = dgates1[s] W_ih1 + dgates0[s+1] W_hh0 ; dgates1[s] was written one launch ago into da1frag[s & 1]
        const bf16x8* a0 = reinterpret_cast<const bf16x8*>(p.da1frag[s & 1]) + (size_t)m_base * 64;
        const bf16x8* a1 = reinterpret_cast<const bf16x8*>(p.da0frag[(s + 1) & 1]) + (size_t)m_base * 64;
        const bf16x8* w0 = reinterpret_cast<const bf16x8*>(p.wTi1frag) + (size_t)jt * nchunk * 64;
        const bf16x8* w1 = reinterpret_cast<const bf16x8*>(p.wT0frag) + (size_t)jt * nchunk * 64;
        if (nchunk == 16 * G) {            // one group per wave and segment: both weight groups in flight up front
            skinny_bf16_dualw<1, G>(a0, a1, w0, w1, wave, 16, lane, acc, issue_epilogue_loads, p.MT);
        } else {
            skinny_bf16<1, G>(a0, w0, nchunk, wave, 16, lane, acc, []() {}, p.MT);
            skinny_bf16<1, G>(a1, w1, nchunk, wave, 16, lane, acc, issue_epilogue_loads, p.MT);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][kg * 4 + r][li] = acc[0][r];
    __syncthreads();
    asm volatile("" ::"v"(ig), "v"(fg), "v"(gg), "v"(og), "v"(c_t), "v"(c_prev), "v"(dyv), "v"(dcc));
    if (!ev) return;
    if (s == 0) c_prev = 0.f;
    const bool active = s < len;

    float da[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        float dh = dyv;
#pragma unroll
        for (int w = 0; w < 16; ++w) dh += red[w][ebl][jl];
        float carry;
        lstm_cell_bwd<true>(dh, dcc, ig, fg, gg, og, c_t, c_prev, da, carry);
        dcar[(size_t)eb * H + eu] = carry;
    }
    float* dg = (L0 ? p.dgx0 : p.dgx1) + ((size_t)s * B + eb) * 4 * H + eu;     // inactive: pad row -> zeros
    unsigned short* dan = L0 ? p.da0frag[s & 1] : p.da1frag[s & 1];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dg[(size_t)g * H] = da[g];
        dan[frag_index(eb, g * H + eu, p.MT)] = f2op16(da[g]);
    }
}

template <int MT, int G, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void lstm2_fwd_step(L2FwdP p) { lstm2_fwd_body<MT, G, NW>(p); }
template <int G>
__global__ __launch_bounds__(1024) void lstm2_bwd_step(L2BwdP p) { lstm2_bwd_body<G>(p); }

// Backward wavefront with a skew of TWO launches.  In lstm2_bwd_body the layer-0 workgroups stream two K = 4H segments
// (dgates1 W_ih1 + dgates0 W_hh0: 2 x 128 KiB of W^T and 2 x 128 KiB of dgates images) while the layer-1 workgroups stream
// one -- and a step is bound by the bytes the busiest CU pulls through its vector-memory path.  The layer-1 workgroup
// already holds the dgates1 image in registers, so it also multiplies it with its W_ih1^T tile and hands the product
// (16 x 16 fp32) to layer 0 through memory; layer 0 picks it up one launch later (hence launch s = layer 1 step s,
// layer 0 step s+2, T+2 launches).  Per workgroup: layer 1 128 + 256 KiB, layer 0 128 + 128 KiB, instead of 256 / 512.
// Needs one group of G chunks per wave (4H/32 == 16 G).  grid = (2 * H/16, MT), 1024 threads.
template <int G>
__global__ __launch_bounds__(1024) void lstm2_bwd_skew2(L2BwdP p) {
    __shared__ float red[16][16][17];
    __shared__ float redp[16][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B, ntile = H >> 4;
    const bool L0 = (int)blockIdx.x >= ntile;
    const int jt = L0 ? blockIdx.x - ntile : blockIdx.x;
    const int s = L0 ? p.s + 2 : p.s;                       // this layer's time step
    // layer 1 also runs one launch past its last step (s = -1): layer 0 step 0 still needs dgates1[0] W_ih1
    if (s >= p.T || s < (L0 ? 0 : -1)) return;
    const bool cellwork = s >= 0;
    const int j0 = jt * 16;
    const int m_base = blockIdx.y, b_base = m_base * 16;
    const int nchunk = (4 * H) >> 5;

    const int role = tid >> 8;                              // 0: cell backward, 1: (layer 1) hand-off writer, 2-3: L2 warm-up of the next rows
    const int rr = tid & 255;
    const int ebl = rr >> 4, eb = b_base + ebl, jl = rr & 15, eu = j0 + jl;
    const bool ev = role == 0 && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    const float* gates = L0 ? p.gates0 : p.gates1;
    const float* cell = L0 ? p.cell0 : p.cell1;
    float* dcar = L0 ? p.dc0 : p.dc1;
    int len;
    float ig, fg, gg, og, c_t, c_prev, dyv, dcc;
    auto issue_epilogue_loads = [&]() {
        len = p.lens[ebc];
        const int t_ld = (role >= 2 && s >= 1) ? s - 1 : (s < 0 ? 0 : s);  // role-less waves: next launch's rows
        int tp = t_ld - 1;
        tp = tp < 0 ? 0 : tp;
        const size_t row = (size_t)t_ld * B + ebc;
        const float* gp = gates + row * 4 * H + eu;
        ig = gp[0]; fg = gp[(size_t)H]; gg = gp[(size_t)2 * H]; og = gp[(size_t)3 * H];
        c_t = cell[row * H + eu];
        c_prev = cell[((size_t)tp * B + ebc) * H + eu];
        // layer 1: external gradient; layer 0: the dgates1[s] W_ih1 product that layer 1 left one launch ago
        dyv = L0 ? p.pbuf[(s + 1) & 1][(size_t)(b_base + ebl) * H + eu] : p.dy1[row * H + eu];
        dcc = dcar[(size_t)ebc * H + eu];
    };

    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accp = {0.f, 0.f, 0.f, 0.f};
    {
        bf16x8 a[G], w0[G], w1[G];
        if (!L0) {
            const bf16x8* af = reinterpret_cast<const bf16x8*>(p.da1frag[(s + 1) & 1]) + (size_t)m_base * 64;   // dgates1[s+1]
            const bf16x8* wa = reinterpret_cast<const bf16x8*>(p.wT1frag) + (size_t)jt * nchunk * 64;          // W_hh1^T
            const bf16x8* wb = reinterpret_cast<const bf16x8*>(p.wTi1frag) + (size_t)jt * nchunk * 64;         // W_ih1^T
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const size_t c = (size_t)(wave + i * 16);
                w0[i] = wa[c * 64 + lane];
                a[i] = af[c * p.MT * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < G; ++i) w1[i] = wb[(size_t)(wave + i * 16) * 64 + lane];
            issue_epilogue_loads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) acc = mfma16(a[i], w0[i], acc);
#pragma unroll
            for (int i = 0; i < G; ++i) accp = mfma16(a[i], w1[i], accp);
        } else {
            const bf16x8* af = reinterpret_cast<const bf16x8*>(p.da0frag[(s + 1) & 1]) + (size_t)m_base * 64;   // dgates0[s+1]
            const bf16x8* wa = reinterpret_cast<const bf16x8*>(p.wT0frag) + (size_t)jt * nchunk * 64;          // W_hh0^T
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const size_t c = (size_t)(wave + i * 16);
                w0[i] = wa[c * 64 + lane];
                a[i] = af[c * p.MT * 64 + lane];
            }
            issue_epilogue_loads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) acc = mfma16(a[i], w0[i], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][kg * 4 + r][li] = acc[r];
        redp[wave][kg * 4 + r][li] = accp[r];
    }
    __syncthreads();
    asm volatile("" ::"v"(ig), "v"(fg), "v"(gg), "v"(og), "v"(c_t), "v"(c_prev), "v"(dyv), "v"(dcc));
    if (role == 1 && !L0) {                                 // hand dgates1[s+1] W_ih1 (for layer 0 step s+1) to the next launch
        float pv = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) pv += redp[w][ebl][jl];
        p.pbuf[s & 1][(size_t)(b_base + ebl) * H + eu] = pv;
        return;
    }
    if (!ev || !cellwork) return;
    if (s == 0) c_prev = 0.f;
    const bool active = s < len;

    float da[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        float dh = dyv;
#pragma unroll
        for (int w = 0; w < 16; ++w) dh += red[w][ebl][jl];
        float carry;
        lstm_cell_bwd<true>(dh, dcc, ig, fg, gg, og, c_t, c_prev, da, carry);
        dcar[(size_t)eb * H + eu] = carry;
    }
    float* dg = (L0 ? p.dgx0 : p.dgx1) + ((size_t)s * B + eb) * 4 * H + eu;     // inactive: pad row -> zeros
    unsigned short* dan = L0 ? p.da0frag[s & 1] : p.da1frag[s & 1];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dg[(size_t)g * H] = da[g];
        dan[frag_index(eb, g * H + eu, p.MT)] = f2op16(da[g]);
    }
}

inline size_t al256(size_t v) { return (v + 255) & ~size_t(255); }
inline int group_of(int per_wave) { return (per_wave % 8 == 0) ? 8 : (per_wave % 4 == 0) ? 4 : (per_wave % 2 == 0) ? 2 : 1; }
inline int mt_of(int B) { return B <= 16 ? 1 : (B <= 32 ? 2 : 4); }

struct Carve {
    char* p;
    template <class T> T* take(size_t bytes) { T* r = reinterpret_cast<T*>(p); p += al256(bytes); return r; }
};

template <int G>
void launch_fwd2(const L2FwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm2_fwd_step<1, G, 4>), grid, dim3(256), 0, st, p);
    else if (mt == 2) hipLaunchKernelGGL((lstm2_fwd_step<2, G, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm2_fwd_step<4, (G > 4 ? 4 : G), 4>), grid, dim3(256), 0, st, p);
}
// 8-wave workgroups: H/32 chunks over 8 waves, GW = chunks per wave and segment
template <int GW>
void launch_fwd2_w8(const L2FwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm2_fwd_step<1, GW, 8>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((lstm2_fwd_step<2, GW, 8>), grid, dim3(512), 0, st, p);
}

// workspace carving shared by the sequence entry points and the fused three-group chain (lstm3.hip)
struct Fwd2Setup { L2FwdP p; size_t state_bytes; unsigned short* w0; unsigned short* w1; };
inline Fwd2Setup setup_fwd2(void* work, int B, int H) {
    const int mt = mt_of(B);
    const size_t BH = (size_t)B * H, frag_act = (size_t)mt * 16 * H * 2, wimg = (size_t)4 * H * H * 2;
    Carve cv{reinterpret_cast<char*>(work)};
    Fwd2Setup u{};
    u.p.c0 = cv.take<float>(BH * 4); u.p.c1 = cv.take<float>(BH * 4);
    u.p.h0frag[0] = cv.take<unsigned short>(frag_act); u.p.h0frag[1] = cv.take<unsigned short>(frag_act);
    u.p.h1frag[0] = cv.take<unsigned short>(frag_act); u.p.h1frag[1] = cv.take<unsigned short>(frag_act);
    u.state_bytes = cv.p - reinterpret_cast<char*>(work);
    u.w0 = cv.take<unsigned short>(wimg);
    u.w1 = cv.take<unsigned short>(2 * wimg);
    u.p.w0frag = u.w0; u.p.w1frag = u.w1;
    return u;
}
struct Bwd2Setup { L2BwdP p; size_t state_bytes; unsigned short *t1, *t0, *ti; };
inline Bwd2Setup setup_bwd2(void* work, int B, int H) {
    const int mt = mt_of(B);
    const size_t BH = (size_t)B * H, frag_act = (size_t)mt * 16 * H * 2, wimg = (size_t)4 * H * H * 2;
    Carve cv{reinterpret_cast<char*>(work)};
    Bwd2Setup u{};
    u.p.dc1 = cv.take<float>(BH * 4); u.p.dc0 = cv.take<float>(BH * 4);
    u.p.da1frag[0] = cv.take<unsigned short>(4 * frag_act); u.p.da1frag[1] = cv.take<unsigned short>(4 * frag_act);
    u.p.da0frag[0] = cv.take<unsigned short>(4 * frag_act); u.p.da0frag[1] = cv.take<unsigned short>(4 * frag_act);
    u.p.pbuf[0] = cv.take<float>((size_t)mt * 16 * H * 4); u.p.pbuf[1] = cv.take<float>((size_t)mt * 16 * H * 4);
    u.state_bytes = cv.p - reinterpret_cast<char*>(work);
    u.t1 = cv.take<unsigned short>(wimg);
    u.t0 = cv.take<unsigned short>(wimg);
    u.ti = cv.take<unsigned short>(wimg);
    u.p.wT1frag = u.t1; u.p.wT0frag = u.t0; u.p.wTi1frag = u.ti;
    u.p.MT = mt;
    return u;
}

}  // namespace

#ifndef FT_LSTM_NO_ENTRY
#if FT_OPFMT == 0
extern "C" int ft_lstm2_supported(int B, int H) { return (B >= 1 && B <= 64 && H >= 128 && H % 128 == 0) ? 1 : 0; }

extern "C" size_t ft_lstm2_workspace_bytes(int B, int H) {
    const int mt = mt_of(B);
    const size_t BH = (size_t)B * H, frag_act = (size_t)mt * 16 * H * 2, wimg = (size_t)4 * H * H * 2;
    const size_t fwd = 2 * al256(BH * 4) + 4 * al256(frag_act) + al256(wimg) + al256(2 * wimg);
    const size_t bwd = 2 * al256(BH * 4) + 4 * al256(4 * frag_act) + 2 * al256((size_t)mt * 16 * H * 4) + 3 * al256(wimg);
    return fwd > bwd ? fwd : bwd;
}
#endif

extern "C" int FT_OPNAME(ft_lstm2_seq_fwd)(const float* gx0, const float* w_hh0, const float* w_ih1, const float* bias1, const float* w_hh1,
                                const int32_t* lens, float* y0, float* gates0, float* cell0, float* y1, float* gates1, float* cell1,
                                void* work, int T, int B, int H, void* stream) {
    FT_CHECK_ARG(gx0 && w_hh0 && w_ih1 && bias1 && w_hh1 && lens && y0 && gates0 && cell0 && y1 && gates1 && cell1 && work);
    FT_CHECK_ARG(T >= 0 && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    if (!ft_lstm2_supported(B, H)) return ft_fail(FT_EUNSUPPORTED, "ft_lstm2_seq_fwd: needs H %% 128 == 0 and B <= 64 (H=%d B=%d)", H, B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int mt = mt_of(B);
    Fwd2Setup u = setup_fwd2(work, B, H);
    L2FwdP& p = u.p;
    unsigned short *w0 = u.w0, *w1 = u.w1;
    const size_t state_bytes = u.state_bytes;
    p.gx0 = gx0; p.bias1 = bias1; p.lens = lens;
    p.y0 = y0; p.gates0 = gates0; p.cell0 = cell0; p.y1 = y1; p.gates1 = gates1; p.cell1 = cell1;
    p.T = T; p.B = B; p.H = H;
    FT_CHECK_HIP(hipMemsetAsync(work, 0, state_bytes, st));
    hipLaunchKernelGGL(make_wfrag_fwd_cat, dim3(2048), dim3(256), 0, st, (const float*)nullptr, 0, w_hh0, H, w0, H);
    hipLaunchKernelGGL(make_wfrag_fwd_cat, dim3(2048), dim3(256), 0, st, w_ih1, H, w_hh1, H, w1, H);
    const int g = group_of((H >> 5) / 4);
    dim3 grid(2 * (H >> 2));
    const int per8 = (H >> 5) / 8;          // chunks per wave with 8 waves (exact when H % 256 == 0)
    const bool w8 = false && (H % 256 == 0) && mt <= 2 && (per8 == 4 || per8 == 2 || per8 == 1);   // measured? no: 128-VGPR budget spills
    static const bool both_off = [] { const char* e = getenv("FT_LSTM2_BOTH"); return e && e[0] == '0'; }();
    const bool both = !both_off && mt <= 2 && (H >> 5) == 32;           // one group of 8 chunks per wave (H = 1024)
    for (int s = 0; s <= T; ++s) {
        p.s = s;
        if (both) {
            if (mt == 1) hipLaunchKernelGGL((lstm2_fwd_both<1, 8>), dim3(H >> 2), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((lstm2_fwd_both<2, 8>), dim3(H >> 2), dim3(256), 0, st, p);
        }
        else if (w8) {
            if (per8 == 4) launch_fwd2_w8<4>(p, mt, grid, st);
            else if (per8 == 2) launch_fwd2_w8<2>(p, mt, grid, st);
            else launch_fwd2_w8<1>(p, mt, grid, st);
        }
        else if (g == 8) launch_fwd2<8>(p, mt, grid, st);
        else if (g == 4) launch_fwd2<4>(p, mt, grid, st);
        else if (g == 2) launch_fwd2<2>(p, mt, grid, st);
        else launch_fwd2<1>(p, mt, grid, st);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm2_seq_bwd)(const float* dy1, const float* w_hh0, const float* w_ih1, const float* w_hh1, const int32_t* lens,
                                const float* gates0, const float* cell0, const float* gates1, const float* cell1,
                                float* dgx0, float* dgx1, void* work, int T, int B, int H, void* stream) {
    FT_CHECK_ARG(dy1 && w_hh0 && w_ih1 && w_hh1 && lens && gates0 && cell0 && gates1 && cell1 && dgx0 && dgx1 && work);
    FT_CHECK_ARG(T >= 0 && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    if (!ft_lstm2_supported(B, H)) return ft_fail(FT_EUNSUPPORTED, "ft_lstm2_seq_bwd: needs H %% 128 == 0 and B <= 64 (H=%d B=%d)", H, B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int mt = mt_of(B);
    Bwd2Setup u = setup_bwd2(work, B, H);
    L2BwdP& p = u.p;
    unsigned short *t1 = u.t1, *t0 = u.t0, *ti = u.ti;
    const size_t state_bytes = u.state_bytes;
    p.dy1 = dy1; p.lens = lens; p.gates1 = gates1; p.cell1 = cell1; p.gates0 = gates0; p.cell0 = cell0;
    p.dgx1 = dgx1; p.dgx0 = dgx0;
    p.T = T; p.B = B; p.H = H;
    FT_CHECK_HIP(hipMemsetAsync(work, 0, state_bytes, st));
    hipLaunchKernelGGL(make_wfrag_bwd_t, dim3(2048), dim3(256), 0, st, w_hh1, t1, H, H);
    hipLaunchKernelGGL(make_wfrag_bwd_t, dim3(2048), dim3(256), 0, st, w_hh0, t0, H, H);
    hipLaunchKernelGGL(make_wfrag_bwd_t, dim3(2048), dim3(256), 0, st, w_ih1, ti, H, H);
    const int g = group_of(((4 * H) >> 5) / 16);
    dim3 grid(2 * (H >> 4), mt);
    static const bool skew_off = [] { const char* e = getenv("FT_LSTM2_SKEW2"); return e && e[0] == '0'; }();
    if (!skew_off && g == 8 && ((4 * H) >> 5) == 16 * 8) {     // one group of 8 chunks per wave (H = 1024): skew-2 hand-off kernel
        for (int s = T - 1; s >= -2; --s) {
            p.s = s;
            hipLaunchKernelGGL(lstm2_bwd_skew2<8>, grid, dim3(1024), 0, st, p);
        }
        FT_CHECK_LAUNCH();
        return FT_OK;
    }
    for (int s = T - 1; s >= -1; --s) {
        p.s = s;
        if (g == 8) hipLaunchKernelGGL(lstm2_bwd_step<8>, grid, dim3(1024), 0, st, p);
        else if (g == 4) hipLaunchKernelGGL(lstm2_bwd_step<4>, grid, dim3(1024), 0, st, p);
        else if (g == 2) hipLaunchKernelGGL(lstm2_bwd_step<2>, grid, dim3(1024), 0, st, p);
        else hipLaunchKernelGGL(lstm2_bwd_step<1>, grid, dim3(1024), 0, st, p);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
#endif  // FT_LSTM_NO_ENTRY
