// Cumulative ("location-sensitive") attention, 16-bit operand modes: ONE fused kernel per frame and direction
// (reference flowtron.py:697-723 `run_cumm_attn_sequence`, :129-152 `AttentionConditioningLayer`, :544-592 `Attention.forward`).
//
// The launch chain of cumm_attn.hip spends a frame on 9 (forward) / 23 (backward) dependent launches of 5-20 us for 1-5 us of
// work each (722 ms per training step at BASELINE configs[1]'s shape).  Here a frame is ONE launch, and everything that does
// not feed the frame-to-frame dependency -- context, dV, the score gradient's dctx . V term, and every WEIGHT gradient -- leaves
// the loop and runs as a few large GEMMs over all frames.
//
// Work split: a workgroup owns a tile of text positions (rows l) of ONE utterance b; all 4 waves of it split the OUTPUT channels.
// Every GEMM is evaluated TRANSPOSED (out^T[channel][l] = W[channel][k] . x[l][k]^T), so that
//   * the weight operand's MFMA fragment (16 channels x 8 consecutive k per lane) is one 16-byte load from a plain row-major
//     16-bit image of the weight in its checkpoint layout ([A][E], [E][NF*K2]) -- streamed from the L2, no LDS staging, each wave
//     reads only ITS channels;
//   * the activation operand (rows l) sits in one small LDS tile shared by the four waves;
//   * a lane ends up with FOUR CONSECUTIVE channels of one row l: float4 loads / stores of text, Q, v, the saved tanh, dtext.
// Forward frame i (grid: ceil(L / 32) x B):
//   softmax of frame i-1's scores (every workgroup of an utterance redoes the <= L-element softmax: cheaper than a hand-off),
//   tile 0 writes attn / logprob / the running sum; location convolution 1 (2 -> 32, k5, VALU) for the tile + halo; convolution
//   2 (32 -> E, k3) + sigmoid as an MFMA GEMM with K = 96; km = text . cond -> LDS; key projection K^T = W_key km^T (MFMA, K = E);
//   t = tanh(Q_i + K) saved (fp32, the only per-frame tensor kept: backward needs 1 - t^2 and t), e = v . t / temperature.
// Backward frame i (grid: ceil(L / 26) x B; a tile COMPUTES 32 rows = 26 own + 3 halo rows either side, because the two
// convolution adjoints reach 1 + 2 rows sideways -- the halo is recomputed instead of exchanged):
//   softmax backward from (dctx_i . V, external gradients, the carried gradients of prev / cumm) -> s_l;
//   dK = s v (1 - t^2) -> LDS + stream; dkm^T = W_key^T dK^T (MFMA, K = A); cond recomputed (MFMA, K = 96);
//   dtext += dkm . cond (own rows, in place); dpre2 = dkm . text . cond (1 - cond) -> LDS + stream; dcol2^T = w2^T dpre2^T
//   (MFMA, K = E, split over the waves); col2im -> dh1 -> relu' -> dpre1; col2im -> gradients of this frame's (cumm, prev)
//   inputs, handed to frame i-1 through a [2][B][L] buffer.
// Streams (16-bit, row = l*B + b inside a frame, frames of a chunk back to back): dK, km, dpre2, col2.  After every chunk of frames:
//   dW_key += dK^T km and dw2 += dpre2^T col2 as split-K image GEMMs over (frames x rows) -- the k-major operand role of
//   ft_gemm_img, nothing is transposed.  dv / db2 / dw1 / db1 accumulate in per-workgroup fp32 slots, summed once at the end.
//
// Compiled twice (FT_OPFMT: bf16 / fp16 operands, entries ftint_cummf_* / ftint_cummf_*_f16); fp32 mode keeps the launch chain.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr float C2 = 2.8853900817779268f;      // 2 log2(e): tanh(x) = 1 - 2 / (2^(C2 x) + 1), as attention.hip
constexpr float L2E = 1.4426950408889634f;
__device__ __forceinline__ float rsig(float x) { return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x) + 1.0f); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-L2E * x)); }

constexpr int NF = 32, K1 = 5, K2 = 3, CK = NF * K2;      // location convolutions 2 -> 32 (k5) -> E (k3); CK = 96 = 3 k-steps
constexpr int XW = 40;                                     // x staging: 38 positions (32 rows + 3 either side) per channel
constexpr int H1P = 33;                                    // h1 staging pitch (34 positions x 32 channels)
constexpr int FWD_ROWS = 32, BWD_OWN = 26, HALO = 3;

__device__ __forceinline__ bf16x8 ld_frag(const unsigned short* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack_op16x2(a, b), pack_op16x2(c, d)); }

// softmax of e[0 .. len) into ps (ps[l] = 0 for len <= l < Lp); red: 8 floats
__device__ __forceinline__ void softmax_block(const float* __restrict__ e, float* ps, float* red, int len, int Lp, int tid) {
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
    for (int l = tid; l < Lp; l += 256) ps[l] = l < len ? ps[l] / s : 0.f;
    __syncthreads();
}

// h1s[jj][c], jj = 0 .. 33 <-> position l = r0 - 1 + jj: relu(b1[c] + sum_{ch,k} w1[c][ch][k] x[ch][l + k - 2]), 0 outside [0, L)
// (the second convolution zero-pads h1 at the ends).  xs[ch][pos] <-> position r0 - 3 + pos.
__device__ __forceinline__ void conv1_h1(const float* xs, float* h1s, const float* __restrict__ w1, const float* __restrict__ b1,
                                          int r0, int L, int tid) {
    const int c = tid & 31;
    float w[2 * K1];
#pragma unroll
    for (int q = 0; q < 2 * K1; ++q) w[q] = w1[c * 2 * K1 + q];
    const float bb = b1[c];
    for (int jj = tid >> 5; jj < 34; jj += 8) {
        const int l = r0 - 1 + jj;
        float h = 0.f;
        if (l >= 0 && l < L) {
            h = bb;
#pragma unroll
            for (int k = 0; k < K1; ++k) { h = fmaf(w[k], xs[jj + k], h); h = fmaf(w[K1 + k], xs[XW + jj + k], h); }
            h = fmaxf(h, 0.f);
        }
        h1s[jj * H1P + c] = h;
    }
}

// MFMA operand of the second convolution for row tile rt: lane (li, kg) holds col2[l = r0 + 16 rt + li][kk = 32 s + 8 kg + 0..7],
// kk = c * 3 + k <-> h1 at position l + k - 1 (w2 is [E][NF][K2]: the same flat order)
__device__ __forceinline__ void col_frags(const float* h1s, int rt, int li, int kg, bf16x8 (&cf)[3]) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float v[8];
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const int kk = 32 * s + 8 * kg + e8, c = kk / 3, k = kk - 3 * c;
            v[e8] = h1s[(16 * rt + li + k) * H1P + c];
        }
        const uint4 u = make_uint4(pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3]), pack_op16x2(v[4], v[5]), pack_op16x2(v[6], v[7]));
        cf[s] = __builtin_bit_cast(bf16x8, u);
    }
}

// pre-activation of cond^T for the e-tile t (rows e = 16 t + 4 kg + r) and the row tile behind cf: acc[r] <-> (e, l = li)
__device__ __forceinline__ f32x4 cond_pre(const unsigned short* __restrict__ w2img, int t, int li, int kg, const bf16x8 (&cf)[3]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* wp = w2img + (size_t)(16 * t + li) * CK + 8 * kg;
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = mfma16(ld_frag(wp + 32 * s), cf[s], acc);
    return acc;
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
struct FwdP {
    const float *text, *Q, *v, *w1, *b1, *b2;
    const unsigned short *w2img, *wkimg;                  // [E][96], [A][E] 16-bit images
    const int* in_lens;
    float *attn, *logprob, *cumm_all, *tsave, *ebuf;
    int T, B, L;
    float inv_temp;
};

template <int NQE, int NQA>
__global__ __launch_bounds__(256, 2) void cummf_fwd_k(FwdP p, int i) {
    constexpr int E = 64 * NQE, A = 64 * NQA, KPE = E + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int T = p.T, B = p.B, L = p.L, Lp = (L + 3) & ~3;
    const int len = min(p.in_lens[b], L);
    const int r0 = j * FWD_ROWS;
    if (r0 >= len && j != 0) return;
    float* ps = reinterpret_cast<float*>(smem);           // [Lp]  attention of frame i-1
    float* xs = ps + Lp;                                  // [2][XW]
    float* h1s = xs + 2 * XW;                             // [34][H1P]
    float* red = h1s + 34 * H1P + 2;                      // [4][32] (first 8 also serve the softmax)
    unsigned short* kmt = reinterpret_cast<unsigned short*>(red + 128);   // [32][KPE]   (offset (Lp + 80 + 1124 + 128) * 4: 16-byte aligned)

    // 1. attention of the previous frame
    if (i > 0) softmax_block(p.ebuf + (size_t)b * L, ps, red, len, Lp, tid);
    else { for (int l = tid; l < Lp; l += 256) ps[l] = 0.f; __syncthreads(); }
    const float* cprev = i > 0 ? p.cumm_all + ((size_t)(i - 1) * B + b) * L : nullptr;
    if (j == 0 && i > 0) {
        const size_t row = ((size_t)b * T + (i - 1)) * L;
        for (int l = tid; l < L; l += 256) {
            const float pl = ps[l];
            p.attn[row + l] = pl;
            p.logprob[row + l] = logf(pl + 1e-8f);
            if (i < T) p.cumm_all[((size_t)i * B + b) * L + l] = cprev[l] + pl;
        }
    }
    if (i >= T || r0 >= len) return;
    // 2. x = [cumm_i ; prev_i] for positions r0 - 3 .. r0 + 34, first convolution
    if (tid < 2 * 38) {
        const int ch = tid / 38, pos = tid - 38 * ch, l = r0 - HALO + pos;
        float x = 0.f;
        if (l >= 0 && l < L && i > 0) x = ch == 0 ? cprev[l] + ps[l] : ps[l];
        xs[ch * XW + pos] = x;
    }
    __syncthreads();
    conv1_h1(xs, h1s, p.w1, p.b1, r0, L, tid);
    __syncthreads();
    // 3. cond = sigmoid(conv2(h1)), km = text . cond -> LDS tile [32 rows][E]
    {
        bf16x8 cf[2][3];
        col_frags(h1s, 0, li, kg, cf[0]);
        col_frags(h1s, 1, li, kg, cf[1]);
#pragma unroll
        for (int q = 0; q < NQE; ++q) {
            const int t = wave + 4 * q, e0 = 16 * t + 4 * kg;
            const float4 bv = *reinterpret_cast<const float4*>(p.b2 + e0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 c = cond_pre(p.w2img, t, li, kg, cf[rt]);
                const int l = r0 + 16 * rt + li;
                float4 tx = make_float4(0.f, 0.f, 0.f, 0.f);
                if (l < L) tx = *reinterpret_cast<const float4*>(p.text + ((size_t)l * B + b) * E + e0);
                *reinterpret_cast<uint2*>(kmt + (16 * rt + li) * KPE + e0) =
                    pack4(tx.x * sigm(c[0] + bv.x), tx.y * sigm(c[1] + bv.y), tx.z * sigm(c[2] + bv.z), tx.w * sigm(c[3] + bv.w));
            }
        }
    }
    __syncthreads();
    // 4. K^T = W_key km^T: wave w owns the a-tiles w, w + 4, ...; both row tiles
    f32x4 acc[NQA][2];
#pragma unroll
    for (int q = 0; q < NQA; ++q) { acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    {
        const unsigned short* wp = p.wkimg + (size_t)(16 * wave + li) * E + 8 * kg;
        const unsigned short* k0 = kmt + li * KPE + 8 * kg;
#pragma unroll 2
        for (int s = 0; s < E / 32; ++s) {
            const bf16x8 b0 = ld_frag(k0 + 32 * s), b1 = ld_frag(k0 + 16 * KPE + 32 * s);
#pragma unroll
            for (int q = 0; q < NQA; ++q) {
                const bf16x8 a = ld_frag(wp + (size_t)q * 64 * E + 32 * s);
                acc[q][0] = mfma16(a, b0, acc[q][0]);
                acc[q][1] = mfma16(a, b1, acc[q][1]);
            }
        }
    }
    // 5. t = tanh(Q_i + K) (saved), e = v . t / temperature
    float part[2] = {0.f, 0.f};
    const size_t RA = (size_t)L * B;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
        const int a0 = 16 * (wave + 4 * q) + 4 * kg;
        const float4 qv = *reinterpret_cast<const float4*>(p.Q + ((size_t)i * B + b) * A + a0);
        const float4 vv = *reinterpret_cast<const float4*>(p.v + a0);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float4 tv;
            tv.x = 1.f - 2.f * rsig(C2 * (qv.x + acc[q][rt][0]));
            tv.y = 1.f - 2.f * rsig(C2 * (qv.y + acc[q][rt][1]));
            tv.z = 1.f - 2.f * rsig(C2 * (qv.z + acc[q][rt][2]));
            tv.w = 1.f - 2.f * rsig(C2 * (qv.w + acc[q][rt][3]));
            part[rt] = fmaf(vv.x, tv.x, fmaf(vv.y, tv.y, fmaf(vv.z, tv.z, fmaf(vv.w, tv.w, part[rt]))));
            const int l = r0 + 16 * rt + li;
            if (l < len) *reinterpret_cast<float4*>(p.tsave + ((size_t)i * RA + (size_t)l * B + b) * A + a0) = tv;
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        part[rt] += __shfl_xor(part[rt], 16, 64);
        part[rt] += __shfl_xor(part[rt], 32, 64);
        if (kg == 0) red[wave * 32 + 16 * rt + li] = part[rt];
    }
    __syncthreads();
    if (tid < 32 && r0 + tid < len)
        p.ebuf[(size_t)b * L + r0 + tid] = ((red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid])) * p.inv_temp;
}

// ---- backward -----------------------------------------------------------------------------------------------------------------
struct BwdP {
    const float *text, *v, *w1, *b1, *b2;
    const unsigned short *w2img, *wkT, *w2T;              // [E][96], [E][A] (= W_key^T), [96][E] (= w2^T) 16-bit images
    const int* in_lens;
    const float *attn, *cumm_all, *tsave, *DV, *dattn, *dlogprob;
    float* gbuf;                                          // [2 parity][2: prev, cumm][B][L]
    float *dQ, *dtext, *dv_part, *db2_part, *dw1_part, *db1_part;
    unsigned short *dK_s, *km_s, *dp2_s, *col2_s;         // streams, [slot][l*B + b][A | E | E | 96]
    int T, B, L;
    float inv_temp;
};

template <int NQE, int NQA>
__global__ __launch_bounds__(256, 1) void cummf_bwd_k(BwdP p, int i, int slot) {
    constexpr int E = 64 * NQE, A = 64 * NQA, KPE = E + 8, KPA = A + 8, DCP = CK + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int T = p.T, B = p.B, L = p.L, Lp = (L + 3) & ~3;
    const int len = min(p.in_lens[b], L);
    const int o0 = BWD_OWN * j, r0 = o0 - HALO;
    if (o0 > len) return;                                 // (row `len` itself still carries a gradient of the first convolution)
    const int wg = b * gridDim.x + j;
    float* ps = reinterpret_cast<float*>(smem);           // [Lp] attention of frame i
    float* ss = ps + Lp;                                  // [Lp] dp, then s
    float* xs = ss + Lp;                                  // [2][XW]
    float* h1s = xs + 2 * XW;                             // [34][H1P]
    float* red = h1s + 34 * H1P + 2;                      // [8]
    float* dqs = red + 8;                                 // [A]
    float* dvs = dqs + A;                                 // [A]
    float* dcs = dvs + A;                                 // [32][DCP]  dcol2
    float* dp1 = dcs + 32 * DCP;                          // [32][H1P]  dpre1 (rows 1 .. 30)
    unsigned short* dkt = reinterpret_cast<unsigned short*>(dp1 + 32 * H1P);      // [32][KPA]
    unsigned short* dpt = dkt + 32 * KPA;                                         // [32][KPE]
    const size_t RA = (size_t)L * B;
    const size_t fr = (size_t)slot * RA;                  // first stream row of this frame
    const float* g_in = p.gbuf + (size_t)((i + 1) & 1) * 2 * B * L;
    float* g_out = p.gbuf + (size_t)(i & 1) * 2 * B * L;

    // 1. softmax backward: s_l = p_l (dp_l - sum_m p_m dp_m) / temperature
    const size_t arow = ((size_t)b * T + i) * L;
    float sum = 0.f;
    for (int l = tid; l < len; l += 256) {
        const float pl = p.attn[arow + l];
        float d = p.DV[arow + l] + g_in[(size_t)b * L + l] + g_in[(size_t)B * L + (size_t)b * L + l];
        if (p.dattn) d += p.dattn[arow + l];
        if (p.dlogprob) d += p.dlogprob[arow + l] / (pl + 1e-8f);
        ps[l] = pl; ss[l] = d;
        sum = fmaf(pl, d, sum);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    for (int q = tid; q < 2 * A; q += 256) dqs[q] = 0.f;               // dqs | dvs
    for (int q = tid; q < 32 * DCP; q += 256) dcs[q] = 0.f;
    // x_i for the positions r0 - 3 .. r0 + 34
    if (tid < 2 * 38) {
        const int ch = tid / 38, pos = tid - 38 * ch, l = r0 - HALO + pos;
        float x = 0.f;
        if (l >= 0 && l < L) x = ch == 0 ? p.cumm_all[((size_t)i * B + b) * L + l] : (i > 0 ? p.attn[arow - L + l] : 0.f);
        xs[ch * XW + pos] = x;
    }
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int l = tid; l < len; l += 256) ss[l] = ps[l] * (ss[l] - sum) * p.inv_temp;
    conv1_h1(xs, h1s, p.w1, p.b1, r0, L, tid);
    __syncthreads();
    // col2 rows of the own positions -> stream (B operand of the dw2 GEMM)
    for (int idx = tid; idx < BWD_OWN * CK; idx += 256) {
        const int jo = idx / CK, ck = idx - CK * jo, l = o0 + jo;
        if (l < len) {
            const int c = ck / 3, k = ck - 3 * c;
            p.col2_s[(fr + (size_t)l * B + b) * CK + ck] = f2op16(h1s[(l - r0 + k) * H1P + c]);
        }
    }
    // 2. dK = s v (1 - t^2) for the 32 computed rows -> LDS tile (B operand of the dkm GEMM) + stream (own rows);
    //    dQ_i = sum_l dK, dv += sum_l s t over the own rows
    {
        const int ag = tid & 31, rg = tid >> 5;
        float4 dq[A / 128], dvp[A / 128], vv[A / 128];
#pragma unroll
        for (int m = 0; m < A / 128; ++m) {
            dq[m] = make_float4(0.f, 0.f, 0.f, 0.f); dvp[m] = dq[m];
            vv[m] = *reinterpret_cast<const float4*>(p.v + 4 * ag + 128 * m);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = rg + 8 * n, l = r0 + row;
            const bool valid = l >= 0 && l < len;
            const float sl = valid ? ss[l] : 0.f;
            const bool own = valid && row >= HALO && row < HALO + BWD_OWN;
#pragma unroll
            for (int m = 0; m < A / 128; ++m) {
                const int a = 4 * ag + 128 * m;
                float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) tv = *reinterpret_cast<const float4*>(p.tsave + ((size_t)i * RA + (size_t)l * B + b) * A + a);
                float4 dk;
                dk.x = sl * vv[m].x * fmaf(-tv.x, tv.x, 1.f);
                dk.y = sl * vv[m].y * fmaf(-tv.y, tv.y, 1.f);
                dk.z = sl * vv[m].z * fmaf(-tv.z, tv.z, 1.f);
                dk.w = sl * vv[m].w * fmaf(-tv.w, tv.w, 1.f);
                const uint2 pk = pack4(dk.x, dk.y, dk.z, dk.w);
                *reinterpret_cast<uint2*>(dkt + row * KPA + a) = pk;
                if (own) {
                    *reinterpret_cast<uint2*>(p.dK_s + (fr + (size_t)l * B + b) * A + a) = pk;
                    dq[m].x += dk.x; dq[m].y += dk.y; dq[m].z += dk.z; dq[m].w += dk.w;
                    dvp[m].x = fmaf(sl, tv.x, dvp[m].x); dvp[m].y = fmaf(sl, tv.y, dvp[m].y);
                    dvp[m].z = fmaf(sl, tv.z, dvp[m].z); dvp[m].w = fmaf(sl, tv.w, dvp[m].w);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < A / 128; ++m) {
            const int a = 4 * ag + 128 * m;
            atomicAdd(dqs + a, dq[m].x); atomicAdd(dqs + a + 1, dq[m].y); atomicAdd(dqs + a + 2, dq[m].z); atomicAdd(dqs + a + 3, dq[m].w);
            atomicAdd(dvs + a, dvp[m].x); atomicAdd(dvs + a + 1, dvp[m].y); atomicAdd(dvs + a + 2, dvp[m].z); atomicAdd(dvs + a + 3, dvp[m].w);
        }
    }
    __syncthreads();
    for (int a = tid; a < A; a += 256) {
        atomicAdd(p.dQ + ((size_t)i * B + b) * A + a, dqs[a]);
        p.dv_part[(size_t)wg * A + a] += dvs[a];
    }
    // 3. dkm^T = W_key^T dK^T: wave w owns the e-tiles w, w + 4, ...
    f32x4 acc[NQE][2];
#pragma unroll
    for (int q = 0; q < NQE; ++q) { acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    {
        const unsigned short* wp = p.wkT + (size_t)(16 * wave + li) * A + 8 * kg;
        const unsigned short* k0 = dkt + li * KPA + 8 * kg;
#pragma unroll 2
        for (int s = 0; s < A / 32; ++s) {
            const bf16x8 b0 = ld_frag(k0 + 32 * s), b1 = ld_frag(k0 + 16 * KPA + 32 * s);
#pragma unroll
            for (int q = 0; q < NQE; ++q) {
                const bf16x8 a = ld_frag(wp + (size_t)q * 64 * A + 32 * s);
                acc[q][0] = mfma16(a, b0, acc[q][0]);
                acc[q][1] = mfma16(a, b1, acc[q][1]);
            }
        }
    }
    // 4. cond again (MFMA, K = 96); km -> stream; dtext += dkm . cond; dpre2 = dkm . text . cond (1 - cond) -> LDS + stream; db2
    {
        bf16x8 cf[2][3];
        col_frags(h1s, 0, li, kg, cf[0]);
        col_frags(h1s, 1, li, kg, cf[1]);
#pragma unroll
        for (int q = 0; q < NQE; ++q) {
            const int t = wave + 4 * q, e0 = 16 * t + 4 * kg;
            const float4 bv = *reinterpret_cast<const float4*>(p.b2 + e0);
            float4 dbv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 cp = cond_pre(p.w2img, t, li, kg, cf[rt]);
                const int row = 16 * rt + li, l = r0 + row;
                const bool inl = l >= 0 && l < len;                    // beyond len: dK = 0, hence dkm = 0
                float4 tx = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inl) tx = *reinterpret_cast<const float4*>(p.text + ((size_t)l * B + b) * E + e0);
                const float c0 = sigm(cp[0] + bv.x), c1 = sigm(cp[1] + bv.y), c2 = sigm(cp[2] + bv.z), c3 = sigm(cp[3] + bv.w);
                const float d0 = acc[q][rt][0], d1 = acc[q][rt][1], d2 = acc[q][rt][2], d3 = acc[q][rt][3];
                float4 dp;
                dp.x = d0 * tx.x * c0 * (1.f - c0); dp.y = d1 * tx.y * c1 * (1.f - c1);
                dp.z = d2 * tx.z * c2 * (1.f - c2); dp.w = d3 * tx.w * c3 * (1.f - c3);
                const uint2 pk = pack4(dp.x, dp.y, dp.z, dp.w);
                *reinterpret_cast<uint2*>(dpt + row * KPE + e0) = pk;
                if (inl && row >= HALO && row < HALO + BWD_OWN) {
                    const size_t g = (fr + (size_t)l * B + b) * E + e0;
                    *reinterpret_cast<uint2*>(p.km_s + g) = pack4(tx.x * c0, tx.y * c1, tx.z * c2, tx.w * c3);
                    *reinterpret_cast<uint2*>(p.dp2_s + g) = pk;
                    float4* dt = reinterpret_cast<float4*>(p.dtext + ((size_t)l * B + b) * E + e0);
                    float4 o = *dt;
                    o.x = fmaf(d0, c0, o.x); o.y = fmaf(d1, c1, o.y); o.z = fmaf(d2, c2, o.z); o.w = fmaf(d3, c3, o.w);
                    *dt = o;
                    dbv.x += dp.x; dbv.y += dp.y; dbv.z += dp.z; dbv.w += dp.w;
                }
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                dbv.x += __shfl_xor(dbv.x, off, 64); dbv.y += __shfl_xor(dbv.y, off, 64);
                dbv.z += __shfl_xor(dbv.z, off, 64); dbv.w += __shfl_xor(dbv.w, off, 64);
            }
            if (li == 0) {
                float4* d = reinterpret_cast<float4*>(p.db2_part + (size_t)wg * E + e0);
                float4 o = *d;
                o.x += dbv.x; o.y += dbv.y; o.z += dbv.z; o.w += dbv.w;
                *d = o;
            }
        }
    }
    __syncthreads();
    // 5. dcol2^T [96][32 rows] = w2^T dpre2^T, the K = E reduction split over the waves, combined by LDS atomics
    {
        f32x4 a3[CK / 16][2];
#pragma unroll
        for (int m = 0; m < CK / 16; ++m) { a3[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; a3[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const unsigned short* wp = p.w2T + (size_t)li * E + 8 * kg;
        const unsigned short* k0 = dpt + li * KPE + 8 * kg;
        for (int s = wave; s < E / 32; s += 4) {
            const bf16x8 b0 = ld_frag(k0 + 32 * s), b1 = ld_frag(k0 + 16 * KPE + 32 * s);
#pragma unroll
            for (int m = 0; m < CK / 16; ++m) {
                const bf16x8 a = ld_frag(wp + (size_t)m * 16 * E + 32 * s);
                a3[m][0] = mfma16(a, b0, a3[m][0]);
                a3[m][1] = mfma16(a, b1, a3[m][1]);
            }
        }
#pragma unroll
        for (int m = 0; m < CK / 16; ++m)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dcs + (16 * rt + li) * DCP + 16 * m + 4 * kg + r, a3[m][rt][r]);
    }
    __syncthreads();
    // 6. dh1[l][c] = sum_k dcol2[l - k + 1][c, k]; dpre1 = dh1 where h1 > 0 (rows 1 .. 30 of the tile)
    for (int idx = tid; idx < 30 * 32; idx += 256) {
        const int jj = 1 + (idx >> 5), c = idx & 31, l = r0 + jj;
        float d = 0.f;
        if (l >= 0 && l < L && h1s[(jj + 1) * H1P + c] > 0.f)
            d = dcs[(jj + 1) * DCP + 3 * c] + dcs[jj * DCP + 3 * c + 1] + dcs[(jj - 1) * DCP + 3 * c + 2];
        dp1[jj * H1P + c] = d;
    }
    __syncthreads();
    // 7. gradients of this frame's inputs (own rows): ds2[l][ch] = sum_{c,k} w1[c][ch][k] dpre1[l - k + 2][c];
    //    prev feeds attn_{i-1} only, cumm every earlier attention
    {
        const int o = tid >> 2, part = tid & 3;
        const int jo = o >> 1, ch = o & 1, jj = HALO + jo;
        float d = 0.f;
        if (o < 2 * BWD_OWN) {
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int c = 8 * part + cc;
#pragma unroll
                for (int k = 0; k < K1; ++k) d = fmaf(p.w1[c * 2 * K1 + ch * K1 + k], dp1[(jj - k + 2) * H1P + c], d);
            }
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        const int l = o0 + jo;
        if (o < 2 * BWD_OWN && part == 0 && l < len) {
            if (ch == 1) g_out[(size_t)b * L + l] = d;
            else g_out[(size_t)B * L + (size_t)b * L + l] = g_in[(size_t)B * L + (size_t)b * L + l] + d;
        }
    }
    // 8. dw1[c][ch][k] += sum_{own l} dpre1[l][c] x[ch][l + k - 2], db1[c] += sum_{own l} dpre1[l][c]
    for (int idx = tid; idx < NF * 2 * K1 + NF; idx += 256) {
        float d = 0.f;
        if (idx < NF * 2 * K1) {
            const int c = idx / (2 * K1), rest = idx - c * 2 * K1, ch = rest / K1, k = rest - ch * K1;
            for (int jj = HALO; jj < HALO + BWD_OWN; ++jj) d = fmaf(dp1[jj * H1P + c], xs[ch * XW + jj + k + 1], d);
            p.dw1_part[(size_t)wg * NF * 2 * K1 + idx] += d;
        } else {
            const int c = idx - NF * 2 * K1;
            for (int jj = HALO; jj < HALO + BWD_OWN; ++jj) d += dp1[jj * H1P + c];
            p.db1_part[(size_t)wg * NF + c] += d;
        }
    }
}

// 16-bit image of a row-major fp32 matrix [rows][cols], optionally transposed: dst [cols][rows]
__global__ void cvt16_k(const float* __restrict__ src, int rows, int cols, unsigned short* __restrict__ dst, int transpose) {
    const long n = (long)rows * cols;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int r = (int)(q / cols), c = (int)(q - (long)r * cols);
        dst[transpose ? (size_t)c * rows + r : (size_t)q] = f2op16(src[q]);
    }
}
// out[c] = sum_w part[w][c]
__global__ void part_sum_k(const float* __restrict__ part, int nw, int n, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += part[(size_t)w * n + c];
    out[c] = s;
}

inline size_t up256(size_t v) { return (v + 255) & ~size_t(255); }

struct Carve {
    unsigned short *w2img, *wkimg, *wkT, *w2T;
    float *ebuf, *gbuf, *DV, *dv_part, *db2_part, *dw1_part, *db1_part;
    unsigned short *dK_s, *km_s, *dp2_s, *col2_s;
    int Tc, nwg;
    size_t part_floats, stream_bytes, total;
};

Carve carve(void* base, int T, int L, int B, int E, int A, bool bwd) {
    Carve c{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = base ? reinterpret_cast<char*>(base) + off : nullptr; off += up256(bytes); return q; };
    c.w2img = reinterpret_cast<unsigned short*>(take((size_t)E * CK * 2 + 256));
    c.wkimg = reinterpret_cast<unsigned short*>(take((size_t)A * E * 2 + 256));
    c.ebuf = reinterpret_cast<float*>(take((size_t)B * L * 4));
    if (bwd) {
        c.wkT = reinterpret_cast<unsigned short*>(take((size_t)A * E * 2 + 256));
        c.w2T = reinterpret_cast<unsigned short*>(take((size_t)E * CK * 2 + 256));
        c.gbuf = reinterpret_cast<float*>(take((size_t)4 * B * L * 4));
        c.DV = reinterpret_cast<float*>(take((size_t)B * T * L * 4));
        c.nwg = B * cdiv(L, BWD_OWN);
        c.part_floats = (size_t)c.nwg * (A + E + NF * 2 * K1 + NF);
        c.dv_part = reinterpret_cast<float*>(take(c.part_floats * 4));
        c.db2_part = c.dv_part ? c.dv_part + (size_t)c.nwg * A : nullptr;
        c.dw1_part = c.dv_part ? c.db2_part + (size_t)c.nwg * E : nullptr;
        c.db1_part = c.dv_part ? c.dw1_part + (size_t)c.nwg * NF * 2 * K1 : nullptr;
        // frames per chunk: ~1.5 GB of streams (the weight-gradient GEMMs run once per chunk)
        const size_t per_frame = (size_t)L * B * (A + 2 * E + CK) * 2;
        long tc = (long)(1500000000ull / per_frame);
        c.Tc = (int)(tc < 4 ? 4 : (tc > T ? T : tc));
        const size_t rows = (size_t)c.Tc * L * B + 288;             // slack: the GEMM tiles read up to 256 columns / 32 rows past the end
        const size_t s0 = off;
        c.dK_s = reinterpret_cast<unsigned short*>(take(rows * A * 2));
        c.km_s = reinterpret_cast<unsigned short*>(take(rows * E * 2));
        c.dp2_s = reinterpret_cast<unsigned short*>(take(rows * E * 2));
        c.col2_s = reinterpret_cast<unsigned short*>(take(rows * CK * 2));
        c.stream_bytes = off - s0;
    }
    c.total = off + 256;
    return c;
}

int bgemm(const float* A, const float* Bm, float* C, int M, int N, int K, long sAm, long sAk, long sBk, long sBn, long ldc, int batch,
          long bsA, long bsB, long bsC, int mode, hipStream_t st) {
    ft_gemm_args a{};
    a.A = A; a.B = Bm; a.C = C; a.bias = nullptr; a.M = M; a.N = N; a.K = K; a.batch = batch;
    a.sAm = sAm; a.sAk = sAk; a.sBk = sBk; a.sBn = sBn; a.ldc = ldc; a.bsA = bsA; a.bsB = bsB; a.bsC = bsC;
    a.alpha = 1.f; a.beta = 0.f; a.act = FT_ACT_NONE; a.mode = mode; a.flags = 0;
    return ft_gemm(&a, st);
}

#define CK_(x) do { int rc_ = (x); if (rc_ != FT_OK) return rc_; } while (0)

}  // namespace

// shapes the fused kernels are instantiated for (config.json: n_text_dim 512 + n_speaker_dim 128 = 640 = n_attn_channels)
int FT_OPNAME(ftint_cummf_supported)(const ft_cumm_attn_args* a) {
    return a->mode == FT_OP16 && a->E == 640 && a->A == 640 && a->NF == NF && a->K1 == K1 && a->K2 == K2 && a->L <= 2048 && a->B <= 65535;
}

size_t FT_OPNAME(ftint_cummf_workspace_bytes)(int T, int L, int B, int E, int A, int backward) {
    return carve(nullptr, T, L, B, E, A, backward != 0).total;
}

int FT_OPNAME(ftint_cummf_fwd)(const ft_cumm_attn_args* a, hipStream_t st) {
    const int T = a->T, B = a->B, L = a->L, E = a->E, A = a->A;
    const Carve c = carve(a->work, T, L, B, E, A, false);
    FT_CHECK_ARG(a->work_bytes >= c.total);
    hipLaunchKernelGGL(cvt16_k, dim3(240), dim3(256), 0, st, a->w2, E, CK, c.w2img, 0);
    hipLaunchKernelGGL(cvt16_k, dim3(1024), dim3(256), 0, st, a->w_key, A, E, c.wkimg, 0);
    FT_CHECK_HIP(hipMemsetAsync(a->cumm_all, 0, sizeof(float) * (size_t)B * L, st));           // cumm_0 = 0
    FwdP p{};
    p.text = a->text; p.Q = a->Q; p.v = a->v; p.w1 = a->w1; p.b1 = a->b1; p.b2 = a->b2; p.w2img = c.w2img; p.wkimg = c.wkimg;
    p.in_lens = a->in_lens; p.attn = a->attn; p.logprob = a->logprob; p.cumm_all = a->cumm_all; p.tsave = a->kproj_all; p.ebuf = c.ebuf;
    p.T = T; p.B = B; p.L = L; p.inv_temp = 1.0f / a->temperature;
    const int Lp = (L + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)Lp + 2 * XW + 34 * H1P + 2 + 128) + (size_t)32 * (E + 8) * 2;
    const dim3 grid(cdiv(L, FWD_ROWS), B);
    for (int i = 0; i <= T; ++i)                     // launch T only closes frame T-1 (softmax, attn, logprob)
        hipLaunchKernelGGL((cummf_fwd_k<10, 10>), grid, dim3(256), lds, st, p, i);
    FT_CHECK_LAUNCH();
    // ctx[t][b][:] = sum_l attn[b][t][l] V[l][b][:]   (not part of the frame-to-frame dependency: one batched GEMM)
    CK_(bgemm(a->attn, a->V, a->ctx, T, A, L, L, 1, (long)B * A, 1, (long)B * A, B, (long)T * L, A, A, a->mode, st));
    return FT_OK;
}

int FT_OPNAME(ftint_cummf_bwd)(const ft_cumm_attn_args* a, const float* dctx, const float* dattn, const float* dlogprob,
                               float* dQ, float* dV, float* dtext, float* dw_key, float* dv, float* dw1, float* db1, float* dw2, float* db2,
                               hipStream_t st) {
    const int T = a->T, B = a->B, L = a->L, E = a->E, A = a->A;
    const Carve c = carve(a->work, T, L, B, E, A, true);
    FT_CHECK_ARG(a->work_bytes >= c.total);
    hipLaunchKernelGGL(cvt16_k, dim3(240), dim3(256), 0, st, a->w2, E, CK, c.w2img, 0);
    hipLaunchKernelGGL(cvt16_k, dim3(240), dim3(256), 0, st, a->w2, E, CK, c.w2T, 1);
    hipLaunchKernelGGL(cvt16_k, dim3(1024), dim3(256), 0, st, a->w_key, A, E, c.wkT, 1);
    FT_CHECK_HIP(hipMemsetAsync(c.gbuf, 0, sizeof(float) * (size_t)4 * B * L, st));
    FT_CHECK_HIP(hipMemsetAsync(c.dv_part, 0, sizeof(float) * c.part_floats, st));
    FT_CHECK_HIP(hipMemsetAsync(c.dK_s, 0, c.stream_bytes, st));          // rows nobody owns (l >= in_len, slack) stay zero for good
    FT_CHECK_HIP(hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)T * B * A, st));
    FT_CHECK_HIP(hipMemsetAsync(dtext, 0, sizeof(float) * (size_t)L * B * E, st));
    // DV[b][t][l] = dctx[t][b] . V[l][b]  and  dV[l][b][:] = sum_t attn[b][t][l] dctx[t][b][:]
    CK_(bgemm(dctx, a->V, c.DV, T, L, A, (long)B * A, 1, 1, (long)B * A, L, B, A, A, (long)T * L, a->mode, st));
    CK_(bgemm(a->attn, dctx, dV, L, A, T, 1, L, (long)B * A, 1, (long)B * A, B, (long)T * L, A, A, a->mode, st));
    BwdP p{};
    p.text = a->text; p.v = a->v; p.w1 = a->w1; p.b1 = a->b1; p.b2 = a->b2; p.w2img = c.w2img; p.wkT = c.wkT; p.w2T = c.w2T;
    p.in_lens = a->in_lens; p.attn = a->attn; p.cumm_all = a->cumm_all; p.tsave = a->kproj_all; p.DV = c.DV; p.dattn = dattn; p.dlogprob = dlogprob;
    p.gbuf = c.gbuf; p.dQ = dQ; p.dtext = dtext; p.dv_part = c.dv_part; p.db2_part = c.db2_part; p.dw1_part = c.dw1_part; p.db1_part = c.db1_part;
    p.dK_s = c.dK_s; p.km_s = c.km_s; p.dp2_s = c.dp2_s; p.col2_s = c.col2_s;
    p.T = T; p.B = B; p.L = L; p.inv_temp = 1.0f / a->temperature;
    const int Lp = (L + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)2 * Lp + 2 * XW + 34 * H1P + 2 + 8 + 2 * A + 32 * (CK + 1) + 32 * H1P) +
                       (size_t)32 * (A + 8) * 2 + (size_t)32 * (E + 8) * 2;
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cummf_bwd_k<10, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(cdiv(L, BWD_OWN), B);
    const size_t RA = (size_t)L * B;
    bool first = true;
    for (int hi = T; hi > 0;) {
        const int lo = ((hi - 1) / c.Tc) * c.Tc, nf = hi - lo;
        for (int i = hi - 1; i >= lo; --i) hipLaunchKernelGGL((cummf_bwd_k<10, 10>), grid, dim3(256), lds, st, p, i, i - lo);
        FT_CHECK_LAUNCH();
        // weight gradients of the chunk: both operands k-major (the reduction runs over frames x rows), split-K
        const long Kr = ((long)nf * (long)RA + 31) / 32 * 32;
        ft_gemm_img_args g{};
        g.alpha = 1.f; g.beta = first ? 0.f : 1.f; g.act = FT_ACT_NONE; g.flags = FT_GEMM_SPLITK; g.a_kmajor = 1; g.b_kmajor = 1;
        g.K = (int)Kr;
        g.A = c.dK_s; g.lda = A; g.B = c.km_s; g.ldb = E; g.C = dw_key; g.ldc = E; g.M = A; g.N = E;
        CK_(FT_OPNAME(ft_gemm_img)(&g, st));
        g.A = c.dp2_s; g.lda = E; g.B = c.col2_s; g.ldb = CK; g.C = dw2; g.ldc = CK; g.M = E; g.N = CK;
        CK_(FT_OPNAME(ft_gemm_img)(&g, st));
        first = false;
        hi = lo;
    }
    hipLaunchKernelGGL(part_sum_k, dim3(cdiv(A, 256)), dim3(256), 0, st, c.dv_part, c.nwg, A, dv);
    hipLaunchKernelGGL(part_sum_k, dim3(cdiv(E, 256)), dim3(256), 0, st, c.db2_part, c.nwg, E, db2);
    hipLaunchKernelGGL(part_sum_k, dim3(cdiv(NF * 2 * K1, 256)), dim3(256), 0, st, c.dw1_part, c.nwg, NF * 2 * K1, dw1);
    hipLaunchKernelGGL(part_sum_k, dim3(1), dim3(256), 0, st, c.db1_part, c.nwg, NF, db1);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
