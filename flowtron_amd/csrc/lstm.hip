// Length-masked LSTM recurrence for training: one kernel launch per time step
// (a dependent kernel boundary is the cheapest chip-wide barrier on gfx950, ~1.5 us, and
// cannot deadlock), each launch spreading the [B,H]x[H,4H] recurrent product over H/4
// workgroups so every CU touches only its own 16 gate rows of W_hh (which stay resident in
// its XCD's L2 across steps: blockIdx -> XCD placement is stable).
//
//   fwd step : block j owns hidden units 4j..4j+3 = 16 gate rows (i,f,g,o x 4).
//              4 waves split K=H; skinny MFMA tiles (M = batch rows, N = 16 gate rows);
//              cross-wave reduce in LDS; the i/f/g/o nonlinearity, cell update,
//              length masking and all stores are fused in the same launch.
//   bwd step : pointwise kernel (dgates from dh, dc) + split-K skinny MFMA
//              dh_rec = dgates . W_hh (against a once-transposed W_hh^T).
// Weight gradients and dx are plain GEMMs over all T*B rows afterwards.
//
// Two operand paths:
//   FT_F32  (parity): fp32 W_hh / h straight from global, v_mfma_f32_16x16x4_f32.
//   FT_BF16 (speed) : once per sequence W_hh (and W_hh^T for backward) is rounded to bf16 and
//              re-laid out in MFMA FRAGMENT ORDER [block][k-chunk][lane][8], and the step kernels
//              keep h_t (resp. dgates_t) in the same fragment order, so every operand load of a
//              step is one fully coalesced 1 KiB global_load_dwordx4 per wave and the per-step L2
//              traffic halves; v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Needs H % 32 == 0.
#include "common.h"
#include "lstm_images.h"

namespace {


// ---- skinny  D[b][n] += sum_k A[b][k] * W[n][k]  over k-chunks c = c0, c0+cs, ... ----
// fp32 : chunk = 16 k, lane (li,kg) loads float4 at k = c*16 + kg*4, 4 x mfma_16x16x4f32
template <int MT>
__device__ __forceinline__ void skinny_f32(const float* const (&arow)[MT], const float* wrow, int K,
                                           int c0, int cs, int kg, f32x4 (&acc)[MT]) {
    const int nchunk = (K + 15) / 16;
    for (int c = c0; c < nchunk; c += cs) {
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
    }
}

// bf16 fragment-order operands: wfrag -> [nchunk][64 lanes][8 bf16], afrag -> [nchunk][MT][64][8].
// Chunks c0, c0+cs, ... ; G chunks are loaded back-to-back (all loads in flight) before their MFMAs issue, so a
// wave pays the L2 round trip once per group instead of once per chunk.  Needs ((nchunk - c0) / cs) % G == 0.
template <int MT, int G, typename Hook>
__device__ __forceinline__ void skinny_bf16(const bf16x8* __restrict__ afrag, const bf16x8* __restrict__ wfrag,
                                            int nchunk, int c0, int cs, int lane, f32x4 (&acc)[MT], Hook&& after_last_loads,
                                            int a_mt = MT) {          // a_mt = m-tiles per k-chunk in the A image (>= MT)
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
    for (int cb = c0 + cs * G; cb < nchunk; cb += cs * G) {      // (H = 1024: a single group, this loop is empty)
        __builtin_amdgcn_sched_barrier(0);
        mfma_group();
        load_group(cb);
    }
    // Straight-line from here: the caller's (HBM-latency) loads are issued BEHIND the last fragment group.  vmcnt
    // retires in order, so no fragment wait below ever includes them, and the compiler can count them exactly.
    after_last_loads();
    __builtin_amdgcn_sched_barrier(0);      // keep every load above in flight: hipcc otherwise sinks them between the MFMAs
    mfma_group();
}

// element (b, k) of a [B, K] activation in fragment order
__device__ __forceinline__ size_t frag_index(int b, int k, int MT) {
    const int c = k >> 5, kg = (k >> 3) & 3, e = k & 7, m = b >> 4, li = b & 15;
    return (((size_t)c * MT + m) * 64 + kg * 16 + li) * 8 + e;
}

struct FwdP {
    const float* gx; const float* w_hh; const int* lens;
    const float* hprev; float* hnext; float* cstate;
    float* y; long ldy; float* gates; float* cell;
    const unsigned short* wfrag;            // bf16 path: [H/4][H/32][64][8]
    const unsigned short* hfrag_prev; unsigned short* hfrag_next;   // bf16 path: [H/32][MT][64][8]
    int s, T, B, H, reverse;
};

template <int MODE, int MT, int G, bool REV>
__device__ __forceinline__ void lstm_fwd_body(const FwdP& p) {
    __shared__ float red[4][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int u0 = blockIdx.x * 4;
    const int H = p.H, B = p.B;

    // Roles.  Threads 0..MT*64-1 own one (batch row, unit) of the cell update; their operands (gx row, cell state)
    // stream from HBM.  With B <= 32 the upper half of the workgroup has no cell role: it touches the gx sectors of
    // step s+1 instead, so that the SAME workgroup (same XCD, same L2) finds them in L2 one launch later.
    // Every thread issues the SAME five loads (addresses clamped in range, no divergent branch): the compiler can then
    // count them, and because vmcnt retires in order and they are issued BEHIND the fragment loads, no fragment wait
    // ever includes their HBM round trip.
    constexpr int NROLE = MT * 64;
    const bool pf_role = (MT <= 2) && tid >= NROLE;
    const int rr = pf_role ? tid - NROLE : tid;
    const int eb = rr >> 2, ul = rr & 3, eu = u0 + ul;
    const bool ev = !pf_role && tid < NROLE && eb < B && eu < H;
    const int ebc = eb < B ? eb : B - 1, euc = eu < H ? eu : H - 1;
    int len;
    float gxv[4], c_old;
    auto issue_epilogue_loads = [&]() {
        len = p.lens[ebc];
        int t_ld;                                 // forward direction: row s (s+1 for the warm-up lanes), independent of len
        if constexpr (!REV) t_ld = (pf_role && p.s + 1 < p.T) ? p.s + 1 : p.s;
        else t_ld = (p.s < len) ? (len - 1 - p.s) : 0;
        const float* gp = p.gx + ((size_t)t_ld * B + ebc) * 4 * H + euc;
#pragma unroll
        for (int g = 0; g < 4; ++g) gxv[g] = gp[(size_t)g * H];
        c_old = p.cstate[(size_t)ebc * H + euc];
    };

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
        issue_epilogue_loads();
        // B operand row for this lane: n = li = gate*4 + ul  ->  W_hh row gate*H + u0 + ul
        const int wu = u0 + (li & 3);
        const float* wrow = (wu < H) ? p.w_hh + (size_t)((li >> 2) * H + wu) * H : nullptr;
        const float* arow[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int b = m * 16 + li;
            arow[m] = (b < B) ? p.hprev + (size_t)b * H : nullptr;
        }
        skinny_f32<MT>(arow, wrow, H, wave, 4, kg, acc);
    } else {
        const int nchunk = H >> 5;
        skinny_bf16<MT, G>(reinterpret_cast<const bf16x8*>(p.hfrag_prev),
                           reinterpret_cast<const bf16x8*>(p.wfrag) + (size_t)blockIdx.x * nchunk * 64, nchunk, wave, 4, lane, acc,
                           issue_epilogue_loads);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + kg * 4 + r][li] = acc[m][r];
    __syncthreads();
    asm volatile("" ::"v"(gxv[0]), "v"(gxv[1]), "v"(gxv[2]), "v"(gxv[3]), "v"(c_old));   // prefetch lanes: keep the loads
    const bool active = p.s < len;
    const int t = REV ? (active ? (len - 1 - p.s) : 0) : p.s;

    if (!ev) return;
    const size_t bu = (size_t)eb * H + eu;
    if (!active) {                               // sample finished: pad row of y is zero, state frozen
        p.y[((size_t)p.s * B + eb) * p.ldy + eu] = 0.f;
        if constexpr (MODE == 0) p.hnext[bu] = p.hprev[bu];
        else p.hfrag_next[frag_index(eb, eu, MT)] = p.hfrag_prev[frag_index(eb, eu, MT)];
        return;
    }
    const size_t row = (size_t)t * B + eb;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = g * 4 + ul;
        pre[g] = red[0][eb][n] + red[1][eb][n] + red[2][eb][n] + red[3][eb][n] + gxv[g];
    }
    float ig, fg, gg, og, c_new, h_new;
    lstm_cell<MODE == 1>(pre, c_old, ig, fg, gg, og, c_new, h_new);
    p.cstate[bu] = c_new;
    if constexpr (MODE == 0) p.hnext[bu] = h_new;
    else p.hfrag_next[frag_index(eb, eu, MT)] = f2op16(h_new);
    p.y[row * p.ldy + eu] = h_new;
    if (p.gates) {
        float* gp = p.gates + row * 4 * H + eu;
        gp[0] = ig; gp[(size_t)H] = fg; gp[(size_t)2 * H] = gg; gp[(size_t)3 * H] = og;
        p.cell[row * H + eu] = c_new;
    }
}

struct BwdP {
    const float* dy; long ldy; const int* lens;
    const float* gates; const float* cell;
    const float* part;      // fp32 path: [4][B][H] partial dh_rec from step s+1
    float* dc_carry;        // [B][H]
    float* da_cur;          // fp32 path: [B][4H]
    float* dgx;             // [T][B][4H]
    const float* wT;        // fp32 path: [H][4H]
    float* part_out;        // fp32 path: [4][B][H]
    const unsigned short* dafrag_prev;  // bf16 path: dgates of step s+1, fragment order over K = 4H: [4H/32][MT][64][8]
    unsigned short* dafrag_next;        // bf16 path: dgates of step s (ping-pong)
    const unsigned short* wTfrag;       // bf16 path: [H/16][4H/32][64][8]
    int s, T, B, H, reverse, MT;        // MT = 16-row batch tiles in the fragment images
};

// ------------------------------------------------------------------ fp32 (parity) path: two launches per step
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
        float carry;
        lstm_cell_bwd<false>(dh, p.dc_carry[idx], ig, fg, gg, og, c_t, c_prev, da, carry);
        p.dc_carry[idx] = carry;
    }
    float* dg = p.dgx + ((size_t)t * B + b) * 4 * H + u;       // inactive: t == s is a pad row -> zeros
    float* dc_ = p.da_cur + (size_t)b * 4 * H + u;
#pragma unroll
    for (int g = 0; g < 4; ++g) { dg[(size_t)g * H] = da[g]; dc_[(size_t)g * H] = da[g]; }
}

// part_out[ks][b][j] = sum_{r in K-slice ks} da_cur[b][r] * wT[j][r]
template <int MT>
__global__ __launch_bounds__(256) void lstm_bwd_matmul(BwdP p) {
    __shared__ float red[4][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B, K4 = 4 * p.H;
    const int j0 = blockIdx.x * 16, ks = blockIdx.y;
    const int KR = K4 / 4;                         // K-slice length (= H), multiple of 4
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int j = j0 + li;
    const float* wrow = (j < H) ? p.wT + (size_t)j * K4 + (size_t)ks * KR : nullptr;
    const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int b = m * 16 + li;
        arow[m] = (b < B) ? p.da_cur + (size_t)b * K4 + (size_t)ks * KR : nullptr;
    }
    skinny_f32<MT>(arow, wrow, KR, wave, 4, kg, acc);
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

// ------------------------------------------------------------------ bf16 path: ONE launch per step
// Block jt owns 16 hidden units for the whole sequence (64 blocks at H = 1024), 16 waves split K = 4H:
//   phase 1: dh_rec[b][j] = sum_r dgates_{s+1}[b][r] * W_hh[r][j]      (fragment-order bf16 operands, full K)
//   phase 2: the LSTM cell backward of step s for the SAME 16 units (they need only this block's dh_rec), writing
//            dgx (fp32, for the batched weight/input-gradient GEMMs) and dgates_s in fragment order for the next launch.
// The gate/cell/dy loads of phase 2 stream from HBM and are issued before phase 1 so their latency hides under it.
// grid = (H/16 unit tiles, MT_total/MT batch tiles): with B = 32 the two 16-row batch halves of a unit tile run as two
// workgroups (same XCD: linear id = y*gridDim.x + x keeps x % 8), each streaming W (128 KB) + HALF of dgates (128 KB)
// instead of one workgroup streaming 384 KB -- the step is bound by bytes per CU.
template <int MT, int G, bool REV>
__device__ __forceinline__ void lstm_bwd_body_bf16(const BwdP& p) {
    __shared__ float red[16][MT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int H = p.H, B = p.B;
    const int j0 = blockIdx.x * 16;
    const int m_base = blockIdx.y * MT, b_base = m_base * 16;

    // Roles as in the forward kernel: threads 0..MT*256-1 own one (batch row, unit) of the cell backward of step s; the
    // remaining waves (B <= 32) run the identical load sequence for step s-1 to warm this XCD's L2 for the next launch.
    constexpr int NROLE = MT * 256;
    const bool pf_role = (MT <= 2) && tid >= NROLE;
    const int rr = pf_role ? tid - NROLE : tid;
    const int ebl = rr >> 4, eb = b_base + ebl, jl = rr & 15, eu = j0 + jl;
    const bool ev = !pf_role && tid < NROLE && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    int len;
    float ig, fg, gg, og, c_t, c_prev, dyv, dcc;
    auto issue_epilogue_loads = [&]() {          // saved gates / cell / dy stream from HBM: issued behind the fragment loads
        len = p.lens[ebc];
        int t_ld;                                 // forward direction: row s (s-1 for the warm-up lanes), independent of len
        if constexpr (!REV) t_ld = (pf_role && p.s >= 1) ? p.s - 1 : p.s;
        else t_ld = (p.s < len) ? (len - 1 - p.s) : p.s;
        int tp = REV ? t_ld + 1 : t_ld - 1;                 // previous step's time index (for c_{t-1})
        tp = tp < 0 ? 0 : (tp > p.T - 1 ? p.T - 1 : tp);
        const size_t row = (size_t)t_ld * B + ebc;
        const float* gp = p.gates + row * 4 * H + eu;
        ig = gp[0]; fg = gp[(size_t)H]; gg = gp[(size_t)2 * H]; og = gp[(size_t)3 * H];
        c_t = p.cell[row * H + eu];
        c_prev = p.cell[((size_t)tp * B + ebc) * H + eu];
        dyv = p.dy[row * p.ldy + eu];
        dcc = p.dc_carry[(size_t)ebc * H + eu];
    };

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const int nchunk = (4 * H) >> 5;
        skinny_bf16<MT, G>(reinterpret_cast<const bf16x8*>(p.dafrag_prev) + (size_t)m_base * 64,
                           reinterpret_cast<const bf16x8*>(p.wTfrag) + (size_t)blockIdx.x * nchunk * 64, nchunk, wave, 16, lane, acc,
                           issue_epilogue_loads, p.MT);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + kg * 4 + r][li] = acc[m][r];
    __syncthreads();
    asm volatile("" ::"v"(ig), "v"(fg), "v"(gg), "v"(og), "v"(c_t), "v"(c_prev), "v"(dyv), "v"(dcc));   // prefetch lanes: keep the loads
    if (!ev) return;
    if (p.s == 0) c_prev = 0.f;
    const bool active = p.s < len;
    int t = REV ? (active ? (len - 1 - p.s) : p.s) : p.s;

    float da[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        float dh = dyv;
#pragma unroll
        for (int w = 0; w < 16; ++w) dh += red[w][ebl][jl];
        float carry;
        lstm_cell_bwd<true>(dh, dcc, ig, fg, gg, og, c_t, c_prev, da, carry);
        p.dc_carry[(size_t)eb * H + eu] = carry;
    }
    if (!active) t = p.s;                                        // inactive: row s is a pad row -> zeros
    float* dg = p.dgx + ((size_t)t * B + eb) * 4 * H + eu;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dg[(size_t)g * H] = da[g];
        p.dafrag_next[frag_index(eb, g * H + eu, p.MT)] = f2op16(da[g]);
    }
}

template <int MODE, int MT, int G, bool REV>
__global__ __launch_bounds__(256) void lstm_fwd_step(FwdP p) { lstm_fwd_body<MODE, MT, G, REV>(p); }
template <int MT, int G, bool REV>
__global__ __launch_bounds__(1024) void lstm_bwd_step_bf16(BwdP p) { lstm_bwd_body_bf16<MT, G, REV>(p); }

// Both directions of a bidirectional layer (the encoder BiLSTM, flowtron.py:488, :505-512) as ONE launch per step: the
// two recurrences are independent, a step is latency-bound, so grid.z = 2 halves the launch count (z = 0 forward in time,
// z = 1 reverse).  bf16 fragment path only.
template <int MT, int G>
__global__ __launch_bounds__(256) void lstm_fwd_pair(FwdP pf, FwdP pr) {
    if (blockIdx.z == 0) lstm_fwd_body<1, MT, G, false>(pf);
    else lstm_fwd_body<1, MT, G, true>(pr);
}
template <int G>
__global__ __launch_bounds__(1024) void lstm_bwd_pair(BwdP pf, BwdP pr) {
    if (blockIdx.z == 0) lstm_bwd_body_bf16<1, G, false>(pf);
    else lstm_bwd_body_bf16<1, G, true>(pr);
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

template <int MODE, int G, bool REV>
void launch_fwd_r(const FwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm_fwd_step<MODE, 1, G, REV>), grid, dim3(256), 0, st, p);
    else if (mt == 2) hipLaunchKernelGGL((lstm_fwd_step<MODE, 2, G, REV>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm_fwd_step<MODE, 4, G, REV>), grid, dim3(256), 0, st, p);
}
template <int MODE, int G>
void launch_fwd_g(const FwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (p.reverse) launch_fwd_r<MODE, G, true>(p, mt, grid, st);
    else launch_fwd_r<MODE, G, false>(p, mt, grid, st);
}
void launch_fwd(const FwdP& p, bool fast, int g, int mt, dim3 grid, hipStream_t st) {
    if (!fast) { launch_fwd_g<0, 1>(p, mt, grid, st); return; }
    if (g == 8 && mt <= 2) launch_fwd_g<1, 8>(p, mt, grid, st);     // 8 chunks x MT<=2 fragments fit the 256-thread VGPR budget
    else if (g >= 4) launch_fwd_g<1, 4>(p, mt, grid, st);
    else if (g == 2) launch_fwd_g<1, 2>(p, mt, grid, st);
    else launch_fwd_g<1, 1>(p, mt, grid, st);
}
template <int G, bool REV>
void launch_bwd_fused_r(const BwdP& p, int mt, dim3 grid, hipStream_t st) {
    // one 16-row batch tile per workgroup: (H/16) x mt workgroups
    hipLaunchKernelGGL((lstm_bwd_step_bf16<1, G, REV>), dim3(grid.x, mt), dim3(1024), 0, st, p);
}
template <int G>
void launch_bwd_fused_g(const BwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (p.reverse) launch_bwd_fused_r<G, true>(p, mt, grid, st);
    else launch_bwd_fused_r<G, false>(p, mt, grid, st);
}
void launch_bwd_fused(const BwdP& p, int g, int mt, dim3 grid, hipStream_t st) {
    if (g == 8) launch_bwd_fused_g<8>(p, mt, grid, st);              // one m-tile per workgroup: 8 x 2 fragments = 64 VGPRs
    else if (g >= 4) launch_bwd_fused_g<4>(p, mt, grid, st);
    else if (g >= 2) launch_bwd_fused_g<2>(p, mt, grid, st);
    else launch_bwd_fused_g<1>(p, mt, grid, st);
}
void launch_bwd_mm(const BwdP& p, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm_bwd_matmul<1>), grid, dim3(256), 0, st, p);
    else if (mt == 2) hipLaunchKernelGGL((lstm_bwd_matmul<2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lstm_bwd_matmul<4>), grid, dim3(256), 0, st, p);
}

inline size_t al256(size_t v) { return (v + 255) & ~size_t(255); }
// fragment path: every wave gets the same number of 32-wide k-chunks (fwd: H/32 over 4 waves, bwd: 4H/32 over 16 waves)
inline bool fast_bf16(int mode, int H) { return mode == FT_OP16 && (H % 128) == 0; }
inline int group_of(int per_wave) { return (per_wave % 8 == 0) ? 8 : (per_wave % 4 == 0) ? 4 : (per_wave % 2 == 0) ? 2 : 1; }

}  // namespace

#ifndef FT_LSTM_NO_ENTRY
#if FT_OPFMT == 0
extern "C" size_t ft_lstm_workspace_bytes(int B, int H) {
    const size_t BH = (size_t)B * H;
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const size_t frag_act = (size_t)mt * 16 * H * 2;                 // one [B_pad, H] bf16 fragment image
    const size_t wfrag = (size_t)4 * H * H * 2;
    const size_t fwd = al256(3 * BH * 4) + al256(2 * frag_act) + al256(wfrag);
    const size_t bwd = al256(9 * BH * 4) + al256((size_t)4 * H * H * 4) + al256(8 * frag_act) + al256(wfrag);
    return fwd > bwd ? fwd : bwd;
}
#endif

extern "C" int FT_OPNAME(ft_lstm_seq_fwd)(const float* gx, const float* w_hh, const int32_t* lens,
                               float* y, int64_t ldy, float* gates, float* cell, void* work,
                               int T, int B, int H, int reverse, int mode, void* stream) {
#if FT_OPFMT == 0
    if (mode == FT_F16) return ft_lstm_seq_fwd_f16(gx, w_hh, lens, y, ldy, gates, cell, work, T, B, H, reverse, mode, stream);
#endif
    FT_CHECK_ARG(gx && w_hh && lens && y && work);
    FT_CHECK_ARG(T >= 0 && B >= 1 && B <= 64 && H >= 4 && H % 4 == 0 && ldy >= H);
    FT_CHECK_ARG((gates == nullptr) == (cell == nullptr));
    FT_CHECK_ARG(mode == FT_F32 || mode == FT_OP16);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(w_hh) % 16 == 0 && reinterpret_cast<uintptr_t>(work) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t BH = (size_t)B * H;
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const bool fast = fast_bf16(mode, H);
    const int g = group_of((H >> 5) / 4);
    char* base = reinterpret_cast<char*>(work);
    float* w = reinterpret_cast<float*>(base);
    float* hbuf[2] = {w, w + BH};
    float* cstate = w + 2 * BH;
    const size_t frag_act = (size_t)mt * 16 * H * 2;
    unsigned short* hfrag[2] = {reinterpret_cast<unsigned short*>(base + al256(3 * BH * 4)),
                                reinterpret_cast<unsigned short*>(base + al256(3 * BH * 4) + frag_act)};
    unsigned short* wfrag = reinterpret_cast<unsigned short*>(base + al256(3 * BH * 4) + al256(2 * frag_act));
    // zero the state, build the fragment image of W_hh
    FT_CHECK_HIP(hipMemsetAsync(base, 0, al256(3 * BH * 4) + al256(2 * frag_act), st));
    if (fast) hipLaunchKernelGGL(make_wfrag_fwd, dim3(2048), dim3(256), 0, st, w_hh, wfrag, H, WfragAux{});
    dim3 grid(cdiv(H, 4));
    for (int s = 0; s < T; ++s) {
        FwdP p{gx, w_hh, lens, hbuf[s & 1], hbuf[(s + 1) & 1], cstate, y, (long)ldy, gates, cell,
               wfrag, hfrag[s & 1], hfrag[(s + 1) & 1], s, T, B, H, reverse};
        launch_fwd(p, fast, g, mt, grid, st);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm_seq_bwd)(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens,
                               const float* gates, const float* cell, float* dgx, void* work,
                               int T, int B, int H, int reverse, int mode, void* stream) {
#if FT_OPFMT == 0
    if (mode == FT_F16) return ft_lstm_seq_bwd_f16(dy, ldy, w_hh, lens, gates, cell, dgx, work, T, B, H, reverse, mode, stream);
#endif
    FT_CHECK_ARG(dy && w_hh && lens && gates && cell && dgx && work);
    FT_CHECK_ARG(T >= 0 && B >= 1 && B <= 64 && H >= 4 && H % 4 == 0 && ldy >= H);
    FT_CHECK_ARG(mode == FT_F32 || mode == FT_OP16);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(work) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t BH = (size_t)B * H;
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const bool fast = fast_bf16(mode, H);
    const int g = group_of(((4 * H) >> 5) / 16);
    char* base = reinterpret_cast<char*>(work);
    float* w = reinterpret_cast<float*>(base);
    float* da_cur = w;                 // [B][4H]
    float* part = w + 4 * BH;          // [4][B][H]
    float* dc_carry = w + 8 * BH;      // [B][H]
    float* wT = reinterpret_cast<float*>(base + al256(9 * BH * 4));            // [H][4H]
    const size_t frag_act = (size_t)mt * 16 * H * 2;
    char* fr = base + al256(9 * BH * 4) + al256((size_t)4 * H * H * 4);
    unsigned short* dafrag[2] = {reinterpret_cast<unsigned short*>(fr), reinterpret_cast<unsigned short*>(fr + 4 * frag_act)};
    unsigned short* wTfrag = reinterpret_cast<unsigned short*>(fr + al256(8 * frag_act));
    // zero the carries, build the W_hh^T images
    FT_CHECK_HIP(hipMemsetAsync(base, 0, al256(9 * BH * 4), st));
    if (fast) {
        FT_CHECK_HIP(hipMemsetAsync(fr, 0, al256(8 * frag_act), st));
        hipLaunchKernelGGL(make_wfrag_bwd, dim3(2048), dim3(256), 0, st, w_hh, wTfrag, H, WfragAux{});
    } else {
        hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(H, 32), cdiv(4 * H, 32)), dim3(32, 8), 0, st, w_hh, wT, 4 * H, H);
    }
    if (fast) {
        dim3 grid(H / 16);
        for (int s = T - 1; s >= 0; --s) {
            BwdP p{dy, (long)ldy, lens, gates, cell, part, dc_carry, da_cur, dgx, wT, part,
                   dafrag[(s + 1) & 1], dafrag[s & 1], wTfrag, s, T, B, H, reverse, mt};
            launch_bwd_fused(p, g, mt, grid, st);
        }
    } else {
        dim3 grid_pw(cdiv((int64_t)B * H, 256)), grid_mm(cdiv(H, 16), 4);
        for (int s = T - 1; s >= 0; --s) {
            BwdP p{dy, (long)ldy, lens, gates, cell, part, dc_carry, da_cur, dgx, wT, part,
                   nullptr, nullptr, nullptr, s, T, B, H, reverse, mt};
            hipLaunchKernelGGL(lstm_bwd_pointwise, grid_pw, dim3(256), 0, st, p);
            if (s > 0) launch_bwd_mm(p, mt, grid_mm, st);
        }
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

#endif  // FT_LSTM_NO_ENTRY

// ---------------------------------------------------------------------------------------------------------------
// bidirectional layer: both directions in one launch per step (lstm_fwd_pair / lstm_bwd_pair)
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct FwdCarve { float* hbuf[2]; float* cstate; unsigned short* hfrag[2]; unsigned short* wfrag; size_t state_bytes; };
FwdCarve carve_fwd(void* work, int B, int H, int mt) {
    const size_t BH = (size_t)B * H, frag_act = (size_t)mt * 16 * H * 2;
    char* base = reinterpret_cast<char*>(work);
    float* w = reinterpret_cast<float*>(base);
    FwdCarve c;
    c.hbuf[0] = w; c.hbuf[1] = w + BH; c.cstate = w + 2 * BH;
    c.hfrag[0] = reinterpret_cast<unsigned short*>(base + al256(3 * BH * 4));
    c.hfrag[1] = reinterpret_cast<unsigned short*>(base + al256(3 * BH * 4) + frag_act);
    c.wfrag = reinterpret_cast<unsigned short*>(base + al256(3 * BH * 4) + al256(2 * frag_act));
    c.state_bytes = al256(3 * BH * 4) + al256(2 * frag_act);
    return c;
}
struct BwdCarve { float* da_cur; float* part; float* dc_carry; float* wT; char* fr; unsigned short* dafrag[2]; unsigned short* wTfrag;
                  size_t carry_bytes, frag_bytes; };
BwdCarve carve_bwd(void* work, int B, int H, int mt) {
    const size_t BH = (size_t)B * H, frag_act = (size_t)mt * 16 * H * 2;
    char* base = reinterpret_cast<char*>(work);
    float* w = reinterpret_cast<float*>(base);
    BwdCarve c;
    c.da_cur = w; c.part = w + 4 * BH; c.dc_carry = w + 8 * BH;
    c.wT = reinterpret_cast<float*>(base + al256(9 * BH * 4));
    c.fr = base + al256(9 * BH * 4) + al256((size_t)4 * H * H * 4);
    c.dafrag[0] = reinterpret_cast<unsigned short*>(c.fr);
    c.dafrag[1] = reinterpret_cast<unsigned short*>(c.fr + 4 * frag_act);
    c.wTfrag = reinterpret_cast<unsigned short*>(c.fr + al256(8 * frag_act));
    c.carry_bytes = al256(9 * BH * 4); c.frag_bytes = al256(8 * frag_act);
    return c;
}
template <int G>
void launch_fwd_pair(const FwdP& pf, const FwdP& pr, int mt, dim3 grid, hipStream_t st) {
    if (mt == 1) hipLaunchKernelGGL((lstm_fwd_pair<1, G>), grid, dim3(256), 0, st, pf, pr);
    else if (mt == 2) hipLaunchKernelGGL((lstm_fwd_pair<2, G>), grid, dim3(256), 0, st, pf, pr);
    else hipLaunchKernelGGL((lstm_fwd_pair<4, (G > 4 ? 4 : G)>), grid, dim3(256), 0, st, pf, pr);
}
}  // namespace

#ifndef FT_LSTM_NO_ENTRY
#if FT_OPFMT == 0
extern "C" int ft_lstm_bidir_supported(int B, int H) { return (B >= 1 && B <= 64 && H >= 128 && H % 128 == 0) ? 1 : 0; }
#endif

extern "C" int FT_OPNAME(ft_lstm_bidir_seq_fwd)(const float* gx_f, const float* gx_r, const float* w_hh_f, const float* w_hh_r,
                                     const int32_t* lens, float* y, int64_t ldy, float* gates_f, float* gates_r,
                                     float* cell_f, float* cell_r, void* work_f, void* work_r, int T, int B, int H, void* stream) {
    FT_CHECK_ARG(gx_f && gx_r && w_hh_f && w_hh_r && lens && y && gates_f && gates_r && cell_f && cell_r && work_f && work_r);
    FT_CHECK_ARG(T >= 0 && ldy >= 2 * (int64_t)H);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(work_f) % 256 == 0 && reinterpret_cast<uintptr_t>(work_r) % 256 == 0);
    if (!ft_lstm_bidir_supported(B, H)) return ft_fail(FT_EUNSUPPORTED, "ft_lstm_bidir_seq_fwd: needs H %% 128 == 0 and B <= 64 (H=%d B=%d)", H, B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const int g = group_of((H >> 5) / 4);
    FwdCarve cf = carve_fwd(work_f, B, H, mt), cr = carve_fwd(work_r, B, H, mt);
    FT_CHECK_HIP(hipMemsetAsync(work_f, 0, cf.state_bytes, st));
    FT_CHECK_HIP(hipMemsetAsync(work_r, 0, cr.state_bytes, st));
    hipLaunchKernelGGL(make_wfrag_fwd, dim3(2048), dim3(256), 0, st, w_hh_f, cf.wfrag, H, WfragAux{});
    hipLaunchKernelGGL(make_wfrag_fwd, dim3(2048), dim3(256), 0, st, w_hh_r, cr.wfrag, H, WfragAux{});
    dim3 grid(H / 4, 1, 2);
    for (int s = 0; s < T; ++s) {
        FwdP pf{gx_f, w_hh_f, lens, cf.hbuf[s & 1], cf.hbuf[(s + 1) & 1], cf.cstate, y, (long)ldy, gates_f, cell_f,
                cf.wfrag, cf.hfrag[s & 1], cf.hfrag[(s + 1) & 1], s, T, B, H, 0};
        FwdP pr{gx_r, w_hh_r, lens, cr.hbuf[s & 1], cr.hbuf[(s + 1) & 1], cr.cstate, y + H, (long)ldy, gates_r, cell_r,
                cr.wfrag, cr.hfrag[s & 1], cr.hfrag[(s + 1) & 1], s, T, B, H, 1};
        if (g == 8 && mt <= 2) launch_fwd_pair<8>(pf, pr, mt, grid, st);
        else if (g >= 4) launch_fwd_pair<4>(pf, pr, mt, grid, st);
        else if (g == 2) launch_fwd_pair<2>(pf, pr, mt, grid, st);
        else launch_fwd_pair<1>(pf, pr, mt, grid, st);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_lstm_bidir_seq_bwd)(const float* dy, int64_t ldy, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                                     const float* gates_f, const float* gates_r, const float* cell_f, const float* cell_r,
                                     float* dgx_f, float* dgx_r, void* work_f, void* work_r, int T, int B, int H, void* stream) {
    FT_CHECK_ARG(dy && w_hh_f && w_hh_r && lens && gates_f && gates_r && cell_f && cell_r && dgx_f && dgx_r && work_f && work_r);
    FT_CHECK_ARG(T >= 0 && ldy >= 2 * (int64_t)H);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(work_f) % 256 == 0 && reinterpret_cast<uintptr_t>(work_r) % 256 == 0);
    if (!ft_lstm_bidir_supported(B, H)) return ft_fail(FT_EUNSUPPORTED, "ft_lstm_bidir_seq_bwd: needs H %% 128 == 0 and B <= 64 (H=%d B=%d)", H, B);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int mt = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const int g = group_of(((4 * H) >> 5) / 16);
    BwdCarve cf = carve_bwd(work_f, B, H, mt), cr = carve_bwd(work_r, B, H, mt);
    FT_CHECK_HIP(hipMemsetAsync(work_f, 0, cf.carry_bytes, st));
    FT_CHECK_HIP(hipMemsetAsync(work_r, 0, cr.carry_bytes, st));
    FT_CHECK_HIP(hipMemsetAsync(cf.fr, 0, cf.frag_bytes, st));
    FT_CHECK_HIP(hipMemsetAsync(cr.fr, 0, cr.frag_bytes, st));
    hipLaunchKernelGGL(make_wfrag_bwd, dim3(2048), dim3(256), 0, st, w_hh_f, cf.wTfrag, H, WfragAux{});
    hipLaunchKernelGGL(make_wfrag_bwd, dim3(2048), dim3(256), 0, st, w_hh_r, cr.wTfrag, H, WfragAux{});
    dim3 grid(H / 16, mt, 2);
    for (int s = T - 1; s >= 0; --s) {
        BwdP pf{dy, (long)ldy, lens, gates_f, cell_f, cf.part, cf.dc_carry, cf.da_cur, dgx_f, cf.wT, cf.part,
                cf.dafrag[(s + 1) & 1], cf.dafrag[s & 1], cf.wTfrag, s, T, B, H, 0, mt};
        BwdP pr{dy + H, (long)ldy, lens, gates_r, cell_r, cr.part, cr.dc_carry, cr.da_cur, dgx_r, cr.wT, cr.part,
                cr.dafrag[(s + 1) & 1], cr.dafrag[s & 1], cr.wTfrag, s, T, B, H, 1, mt};
        if (g == 8) hipLaunchKernelGGL(lstm_bwd_pair<8>, grid, dim3(1024), 0, st, pf, pr);
        else if (g >= 4) hipLaunchKernelGGL(lstm_bwd_pair<4>, grid, dim3(1024), 0, st, pf, pr);
        else if (g >= 2) hipLaunchKernelGGL(lstm_bwd_pair<2>, grid, dim3(1024), 0, st, pf, pr);
        else hipLaunchKernelGGL(lstm_bwd_pair<1>, grid, dim3(1024), 0, st, pf, pr);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
#endif  // FT_LSTM_NO_ENTRY
