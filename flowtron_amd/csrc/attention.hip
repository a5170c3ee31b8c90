// Additive (Bahdanau) attention of the Flowtron AR step, reference flowtron.py:544-583.
//
//   e[b,t,l] = (1/temp) * sum_a v[a] * tanh(Q[t,b,a] + K[l,b,a])      l < in_lens[b]
//   p = softmax_l(e);  prior: u = log(p+1e-20)+log(prior+1e-20), attn = softmax_l(u)
//
// The reference materialises the B*T*L*A tanh tensor (4.85 GB in fp16 at B=32,T=800,
// L=148) and keeps it for backward.  Here it never leaves the register file:
//   forward  : one workgroup per (b, 32 query rows); Q and K a-chunks are staged in LDS,
//              every thread owns a 4x4 (t,l) micro-tile so each LDS float4 feeds 16 tanh;
//              scores land in an LDS [32][L] tile where the softmax / prior posterior is
//              finished by one wave per 8 rows.  Transcendental-ALU bound.
//   backward : softmax/posterior row kernel -> de, then ONE kernel (workgroup per (b, 32 t, 256 a),
//              lane <-> a) that recomputes tanh once and produces dQ (plain stores), dK (one fp32
//              atomic per (l, a) per 32-row tile) and dv.
#include "common.h"

namespace {

constexpr int TT = 32;       // query rows per workgroup
constexpr int LT = 128;      // key columns per pass
constexpr int AC = 64;       // a-chunk staged per iteration
constexpr int LDA = AC + 4;  // LDS row stride (floats): 16-lane groups of ds_read_b128 hit 64 distinct banks

// tanh(x) = 1 - 2/(exp(2x)+1) with the hardware exp/rcp (|abs err| ~ 2e-7, exact limits at +-inf), and the argument
// scale folded into the operands once: with q' = C2*q, k' = C2*k
// (C2 = 2*log2(e)),  r = 1 / (2^(q'+k') + 1),  tanh = 1 - 2r,  1 - tanh^2 = 4 r (1 - r): one add, v_exp_f32, one add,
// v_rcp_f32 per element, and sum_a v[a]*tanh = sum_a v[a] - 2 sum_a v[a]*r keeps a single FMA in the forward loop.
constexpr float C2 = 2.8853900817779268f;
__device__ __forceinline__ float rsig(float x) { return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x) + 1.0f); }
// PRODUCT form of the same quantity: 2^(q'+k') = 2^q' 2^k', so with Eq = 2^q' (once per query element) and Ek = 2^k' (once per key
// element) r = 1 / (Eq Ek + 1) costs ONE transcendental (v_rcp_f32) per (t, l, a) instead of two.  Used only while every staged
// |q'|, |k'| <= EXP_SAFE (|q|, |k| <= 20.8: both factors and their product stay finite and normal); a tile that holds a larger
// value -- decided on the data -- takes the sum form above, so the result never depends on a range assumption.
constexpr float EXP_SAFE = 60.0f;
__device__ __forceinline__ float rsig_prod(float eq, float ek) { return __builtin_amdgcn_rcpf(fmaf(eq, ek, 1.0f)); }
template <bool PROD>
__device__ __forceinline__ float rs(float q, float k) {
    if constexpr (PROD) return rsig_prod(q, k);
    else return rsig(q + k);
}

// scores of one a-chunk for a 4x4 (t, l) micro-tile (UNI: one query row only); qs / ks hold q', k' (sum form) or 2^q', 2^k' (PROD).
// PROD form with PACKED fp32 math (round 6): the two non-transcendental operations per element -- Eq Ek + 1 and acc += v r -- run as
// v_pk_fma_f32 on the (x, y) / (z, w) halves of the staged float4s, the accumulators as (even a, odd a) pairs that are added once at
// the end of the chunk: 2 packed instructions per 4 elements instead of 8 scalar ones beside the 4 v_rcp_f32 (0.556 -> 0.529 ms per
// call at the bench shape: the pass is stall-bound at two waves per SIMD -- 239 VGPRs, 64 KB of LDS --, VALU busy 44 %,
// profiles/r06_pmc_VALU_TRANS.json; the backward pass, VALU busy 71 %, gains more from the same change).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <bool PROD, bool UNI>
__device__ __forceinline__ void score_chunk(const float* qs, const float* ks, const float* vs, int tg, int tl, const bool (&jact)[4],
                                            float (&acc)[4][4], float& vsum, int LDAs, int ACs) {
    if constexpr (PROD) {
        f32x2 acc2[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x2){0.f, 0.f};
        const f32x2 one = {1.f, 1.f};
#pragma unroll 2
        for (int a4 = 0; a4 < ACs / 4; ++a4) {
            const float4 vv = *reinterpret_cast<const float4*>(vs + a4 * 4);
            vsum += (vv.x + vv.y) + (vv.z + vv.w);
            const f32x2 v01 = {vv.x, vv.y}, v23 = {vv.z, vv.w};
            float4 q[4];
#pragma unroll
            for (int i = 0; i < (UNI ? 1 : 4); ++i) q[i] = *reinterpret_cast<const float4*>(qs + ((UNI ? 0 : tg * 4) + i) * LDAs + a4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (jact[j]) {
                    const float4 kk = *reinterpret_cast<const float4*>(ks + (j * 32 + tl) * LDAs + a4 * 4);
                    const f32x2 k01 = {kk.x, kk.y}, k23 = {kk.z, kk.w};
#pragma unroll
                    for (int i = 0; i < (UNI ? 1 : 4); ++i) {
                        const f32x2 d01 = pk_fma((f32x2){q[i].x, q[i].y}, k01, one), d23 = pk_fma((f32x2){q[i].z, q[i].w}, k23, one);
                        const f32x2 r01 = {__builtin_amdgcn_rcpf(d01.x), __builtin_amdgcn_rcpf(d01.y)};
                        const f32x2 r23 = {__builtin_amdgcn_rcpf(d23.x), __builtin_amdgcn_rcpf(d23.y)};
                        acc2[i][j] = pk_fma(v23, r23, pk_fma(v01, r01, acc2[i][j]));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (UNI ? 1 : 4); ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += acc2[i][j].x + acc2[i][j].y;
        return;
    }
#pragma unroll 2
    for (int a4 = 0; a4 < ACs / 4; ++a4) {
        const float4 vv = *reinterpret_cast<const float4*>(vs + a4 * 4);
        vsum += (vv.x + vv.y) + (vv.z + vv.w);
        float4 q[4];
#pragma unroll
        for (int i = 0; i < (UNI ? 1 : 4); ++i) q[i] = *reinterpret_cast<const float4*>(qs + ((UNI ? 0 : tg * 4) + i) * LDAs + a4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (jact[j]) {
                const float4 kk = *reinterpret_cast<const float4*>(ks + (j * 32 + tl) * LDAs + a4 * 4);
#pragma unroll
                for (int i = 0; i < (UNI ? 1 : 4); ++i) {
                    acc[i][j] += vv.x * rs<PROD>(q[i].x, kk.x) + vv.y * rs<PROD>(q[i].y, kk.y) +
                                 vv.z * rs<PROD>(q[i].z, kk.z) + vv.w * rs<PROD>(q[i].w, kk.w);
                }
            }
        }
    }
}

template <bool HAS_PRIOR>
__global__ __launch_bounds__(256) void attn_fwd_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                  const float* __restrict__ v, const int* __restrict__ in_lens,
                                                  const float* __restrict__ prior, float* __restrict__ attn,
                                                  float* __restrict__ logprob, float* __restrict__ p_save,
                                                  int T, int B, int L, int A, int LP, float inv_temp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qs = smem;                 // [TT][LDA]
    float* ks = qs + TT * LDA;        // [LT][LDA]
    float* vs = ks + LT * LDA;        // [AC]
    float* es = vs + AC;              // [TT][LP]

    const int tid = threadIdx.x;
    const int tl = tid & 31, tg = tid >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = min(in_lens[b], L);
    const int nlt = (len + LT - 1) / LT;

    // Tiles of padded frames (~30 % of a batch's rows): every query row of such a tile is the projection of the same zero LSTM
    // output, so all TT score rows are equal.  Decided on the DATA, not on lengths (exact for any input): if the TT query rows
    // of this tile are bit-identical, the scores are evaluated for ONE row (by the 32 threads of row group 0) and copied; the
    // per-row softmax / prior posterior below still runs for every row (the prior may differ from frame to frame).
    bool differs = false;
    {
        const float* q0 = Q + ((long)t0 * B + b) * A;
        for (int idx = tid; idx < TT * AC && !differs; idx += 256) {            // cheap pre-filter: the first a-chunk
            const int row = idx >> 6, col = idx & 63;
            if (t0 + row < T && col < A) differs = Q[((long)(t0 + row) * B + b) * A + col] != q0[col];
        }
        if (!__syncthreads_or(differs ? 1 : 0)) {
            for (int idx = tid; idx < TT * A; idx += 256) {
                const int row = idx / A, col = idx % A;
                if (t0 + row < T && Q[((long)(t0 + row) * B + b) * A + col] != q0[col]) { differs = true; break; }
            }
        }
    }
    const bool uniform_q = !__syncthreads_or(differs ? 1 : 0);
    const bool idle = uniform_q && tg != 0;              // row groups 1..7 have nothing to compute in a uniform tile

    for (int lt = 0; lt < nlt; ++lt) {
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        bool jact[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) jact[j] = (lt * LT + j * 32) < len;
        float vsum = 0.f;

        for (int a0 = 0; a0 < A; a0 += AC) {
            // load the chunk into registers first (no LDS touched yet), decide the form on the data, then stage
            float qv[(TT * AC) / 256], kv[(LT * AC) / 256];
            bool big = false;
#pragma unroll
            for (int r = 0; r < (TT * AC) / 256; ++r) {
                const int idx = tid + 256 * r;
                const int row = idx >> 6, col = idx & 63;
                const int t = t0 + row, a = a0 + col;
                qv[r] = (t < T && a < A) ? C2 * Q[((long)t * B + b) * A + a] : 0.f;
                big |= !(fabsf(qv[r]) <= EXP_SAFE);
            }
#pragma unroll
            for (int r = 0; r < (LT * AC) / 256; ++r) {
                const int idx = tid + 256 * r;
                const int row = idx >> 6, col = idx & 63;
                const int l = lt * LT + row, a = a0 + col;
                kv[r] = (l < len && a < A) ? C2 * K[((long)l * B + b) * A + a] : 0.f;
                big |= !(fabsf(kv[r]) <= EXP_SAFE);
            }
            const bool prod = !__syncthreads_or(big ? 1 : 0);      // (also: the previous chunk's LDS reads are done)
#pragma unroll
            for (int r = 0; r < (TT * AC) / 256; ++r) {
                const int idx = tid + 256 * r;
                qs[(idx >> 6) * LDA + (idx & 63)] = prod ? __builtin_amdgcn_exp2f(qv[r]) : qv[r];
            }
#pragma unroll
            for (int r = 0; r < (LT * AC) / 256; ++r) {
                const int idx = tid + 256 * r;
                ks[(idx >> 6) * LDA + (idx & 63)] = prod ? __builtin_amdgcn_exp2f(kv[r]) : kv[r];
            }
            if (tid < AC) vs[tid] = (a0 + tid < A) ? v[a0 + tid] : 0.f;
            __syncthreads();
            if (uniform_q) {                                   // one score row (same summation order as the general path)
                if (!idle) {
                    if (prod) score_chunk<true, true>(qs, ks, vs, tg, tl, jact, acc, vsum, LDA, AC);
                    else score_chunk<false, true>(qs, ks, vs, tg, tl, jact, acc, vsum, LDA, AC);
                }
                continue;
            }
            if (prod) score_chunk<true, false>(qs, ks, vs, tg, tl, jact, acc, vsum, LDA, AC);
            else score_chunk<false, false>(qs, ks, vs, tg, tl, jact, acc, vsum, LDA, AC);
        }
        if (uniform_q) {
            if (!idle) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (jact[j]) {
                        const float e = (vsum - 2.f * acc[0][j]) * inv_temp;
                        for (int r = 0; r < TT; ++r) es[r * LP + lt * LT + j * 32 + tl] = e;
                    }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (jact[j]) es[(tg * 4 + i) * LP + lt * LT + j * 32 + tl] = (vsum - 2.f * acc[i][j]) * inv_temp;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < TT / 4; ++r) {
        const int row = wave * (TT / 4) + r;
        const int t = t0 + row;
        if (t >= T) break;
        float* e = es + row * LP;
        float m = -INFINITY;
        for (int l = lane; l < len; l += 64) m = fmaxf(m, e[l]);
        m = wave_max(m);
        float s = 0.f;
        for (int l = lane; l < len; l += 64) { const float p = expf(e[l] - m); e[l] = p; s += p; }
        s = wave_sum(s);
        const long base = ((long)b * T + t) * L;
        if constexpr (!HAS_PRIOR) {
            for (int l = lane; l < L; l += 64) {
                const float p = (l < len) ? e[l] / s : 0.f;
                attn[base + l] = p;
                logprob[base + l] = logf(p + 1e-8f);
            }
        } else {
            float m2 = -INFINITY;
            for (int l = lane; l < L; l += 64) {
                const float p = (l < len) ? e[l] / s : 0.f;
                const float u = logf(p + 1e-20f) + logf(prior[base + l] + 1e-20f);
                logprob[base + l] = u;
                p_save[base + l] = p;
                if (l < len) { e[l] = u; m2 = fmaxf(m2, u); }
            }
            m2 = wave_max(m2);
            float s2 = 0.f;
            for (int l = lane; l < len; l += 64) { const float p = expf(e[l] - m2); e[l] = p; s2 += p; }
            s2 = wave_sum(s2);
            for (int l = lane; l < L; l += 64) attn[base + l] = (l < len) ? e[l] / s2 : 0.f;
        }
    }
}

// one wave per (b,t) row: de = d(loss)/d(e) * (1/temp), 0 at l >= in_lens[b]
template <bool HAS_PRIOR>
__global__ __launch_bounds__(256) void attn_softmax_bwd_k(const float* __restrict__ attn, const float* __restrict__ p_save,
                                                          const float* __restrict__ dattn, const float* __restrict__ dlogprob,
                                                          const int* __restrict__ in_lens, float* __restrict__ de,
                                                          int T, int B, int L, float inv_temp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)B * T) return;
    const int b = (int)(row / T);
    const int len = min(in_lens[b], L);
    const long base = row * L;
    float s2 = 0.f;
    if constexpr (HAS_PRIOR) {
        float s1 = 0.f;
        for (int l = lane; l < len; l += 64) s1 += attn[base + l] * dattn[base + l];
        s1 = wave_sum(s1);
        for (int l = lane; l < len; l += 64) {
            const float p = p_save[base + l];
            float du = attn[base + l] * (dattn[base + l] - s1);
            if (dlogprob) du += dlogprob[base + l];
            s2 += p * (du / (p + 1e-20f));
        }
        s2 = wave_sum(s2);
        for (int l = lane; l < L; l += 64) {
            float o = 0.f;
            if (l < len) {
                const float p = p_save[base + l];
                float du = attn[base + l] * (dattn[base + l] - s1);
                if (dlogprob) du += dlogprob[base + l];
                o = p * (du / (p + 1e-20f) - s2) * inv_temp;
            }
            de[base + l] = o;
        }
    } else {
        for (int l = lane; l < len; l += 64) {
            const float p = attn[base + l];
            float dp = dattn[base + l];
            if (dlogprob) dp += dlogprob[base + l] / (p + 1e-8f);
            s2 += p * dp;
        }
        s2 = wave_sum(s2);
        for (int l = lane; l < L; l += 64) {
            float o = 0.f;
            if (l < len) {
                const float p = attn[base + l];
                float dp = dattn[base + l];
                if (dlogprob) dp += dlogprob[base + l] / (p + 1e-8f);
                o = p * (dp - s2) * inv_temp;
            }
            de[base + l] = o;
        }
    }
}

// dQ, dK and dv from ONE evaluation of the tanh tensor.  grid = (ceil(T/32), B, a-chunk groups); lane <-> a (one 64-wide
// a-chunk per wave, 1-5 waves per workgroup chosen by the launcher), every thread keeps 32 query rows in registers and walks the keys:
//   g[t,l,a] = de[b,t,l] * r(1-r),  dQ[t,a] = 4 v[a] sum_l g   (owned by one thread: plain store),
//   dK[l,a] += 4 v[a] sum_{t in tile} g   (one fp32 atomic per (l, a) per 32-row tile; dK must be zeroed),
//   dv[a]  += sum_{t,l} de * tanh.
__global__ __launch_bounds__(256) void attn_dqdk_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                   const float* __restrict__ v, const int* __restrict__ in_lens,
                                                   const float* __restrict__ de, float* __restrict__ dQ, float* __restrict__ dK,
                                                   float* __restrict__ dv, int T, int B, int L, int A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* de_s = smem;                 // [len][32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int len = min(in_lens[b], L);
    int nz = 0;
    for (int idx = tid; idx < len * 32; idx += (int)blockDim.x) {
        const int r = idx & 31, l = idx >> 5;
        const float v = (t0 + r < T) ? de[((long)b * T + t0 + r) * L + l] : 0.f;
        de_s[idx] = v;
        nz |= (v != 0.f);
    }
    const int aw = blockIdx.z * (int)blockDim.x + w * 64;
    if (!__syncthreads_or(nz)) {
        // the whole 32-row tile carries no gradient (padded frames of a short utterance: ~30 % of the rows of a batch):
        // dQ = 0, no contribution to dK / dv -- skip the tanh recomputation.  Exact: decided on the data, not on lengths.
        const int a = aw + lane;
        if (a < A)
            for (int i = 0; i < 32; ++i)
                if (t0 + i < T) dQ[((long)(t0 + i) * B + b) * A + a] = 0.f;
        return;
    }
    if (aw >= A) return;                // wave-uniform
    const int a = aw + lane;
    const bool av = a < A;
    const int ac = av ? a : A - 1;      // clamp: keeps the loads unconditional, results of idle lanes are dropped
    // rows in PAIRS (2 i, 2 i + 1): the non-transcendental arithmetic of an element -- Eq Ek + 1, u = r - r^2, dq += d u, dk += d u,
    // dv' += d r -- runs as v_pk_fma_f32 (round 6: five packed instructions per two elements where six scalar ones per element stood;
    // sum_l,t d tanh = sum d - 2 sum d r, with sum d taken once per key from the de tile)
    f32x2 eq2[16], dq2[16];
    bool qbig = false;
    auto qrow = [&](int i) { return C2 * Q[((long)min(t0 + i, T - 1) * B + b) * A + ac]; };     // (the rare sum-form path re-reads it: 32 registers less)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float q0 = qrow(2 * i), q1 = qrow(2 * i + 1);
        qbig |= !(fabsf(q0) <= EXP_SAFE) | !(fabsf(q1) <= EXP_SAFE);
        eq2[i] = (f32x2){__builtin_amdgcn_exp2f(q0), __builtin_amdgcn_exp2f(q1)};
        dq2[i] = (f32x2){0.f, 0.f};
    }
    float dva = 0.f;
    const float* kp = K + (long)b * A + ac;
    const long ks = (long)B * A;
    const float va4 = 4.f * v[ac];
    float kv_next = (len > 0) ? C2 * kp[0] : 0.f;
    const f32x2 one = {1.f, 1.f};
    for (int l = 0; l < len; ++l) {
        const float kv = kv_next;
        if (l + 1 < len) kv_next = C2 * kp[(long)(l + 1) * ks];
        float dkp = 0.f;
        // product form (one v_rcp per element) unless this wave holds an out-of-range value for this key (wave-uniform choice)
        if (!__any(qbig || !(fabsf(kv) <= EXP_SAFE))) {
            const float ek = __builtin_amdgcn_exp2f(kv);
            const f32x2 ek2 = {ek, ek};
            f32x2 dk2 = {0.f, 0.f}, dr2 = {0.f, 0.f}, ds2 = {0.f, 0.f};
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                const float4 d4 = *reinterpret_cast<const float4*>(de_s + l * 32 + i4 * 4);     // same address in every lane: broadcast
                const f32x2 d01 = {d4.x, d4.y}, d23 = {d4.z, d4.w};
                ds2 = ds2 + d01 + d23;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = i4 * 2 + h;
                    const f32x2 d = h ? d23 : d01;
                    const f32x2 den = pk_fma(eq2[i], ek2, one);
                    const f32x2 r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                    const f32x2 u = pk_fma(-r, r, r);            // r (1 - r) = (1 - tanh^2) / 4
                    dq2[i] = pk_fma(d, u, dq2[i]);
                    dk2 = pk_fma(d, u, dk2);
                    dr2 = pk_fma(d, r, dr2);
                }
            }
            dkp = dk2.x + dk2.y;
            dva += (ds2.x + ds2.y) - 2.f * (dr2.x + dr2.y);
        } else {
#pragma unroll
            for (int i4 = 0; i4 < 8; ++i4) {
                const float4 d4 = *reinterpret_cast<const float4*>(de_s + l * 32 + i4 * 4);
                const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i4 * 4 + j;
                    const float r = rsig(qrow(i) + kv);
                    const float u = fmaf(-r, r, r);
                    if (j & 1) dq2[i >> 1].y = fmaf(d[j], u, dq2[i >> 1].y);
                    else dq2[i >> 1].x = fmaf(d[j], u, dq2[i >> 1].x);
                    dkp = fmaf(d[j], u, dkp);
                    dva = fmaf(d[j], fmaf(-2.f, r, 1.f), dva);
                }
            }
        }
        if (av) atomicAdd(dK + ((long)l * B + b) * A + a, dkp * va4);
    }
    if (av) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = t0 + i;
            if (t < T) dQ[((long)t * B + b) * A + a] = ((i & 1) ? dq2[i >> 1].y : dq2[i >> 1].x) * va4;
        }
        atomicAdd(dv + a, dva);
    }
}

constexpr int MAX_LDS = 160 * 1024;

}  // namespace

extern "C" int ft_attention_fwd(const float* Q, const float* K, const float* v, const int32_t* in_lens, const float* prior,
                                float* attn, float* logprob, float* p_save,
                                int T, int B, int L, int A, float temperature, void* stream) {
    FT_CHECK_ARG(Q && K && v && in_lens && attn && logprob);
    FT_CHECK_ARG(T >= 1 && B >= 1 && L >= 1 && A >= 1 && temperature > 0.f);
    FT_CHECK_ARG(prior == nullptr || p_save != nullptr);
    FT_CHECK_ARG(B <= 65535);
    const int LP = cdiv(L, LT) * LT;
    const size_t lds = sizeof(float) * ((size_t)TT * LDA + (size_t)LT * LDA + AC + (size_t)TT * LP);
    if (lds > (size_t)MAX_LDS - 1024) return ft_fail(FT_EUNSUPPORTED, "ft_attention_fwd: L=%d needs %zu B of LDS (max %d)", L, lds, MAX_LDS - 1024);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv(T, TT), B);
    // (the kernel also has a few static LDS bytes -- __syncthreads_or -- so ask for what this launch needs, not for all 160 KiB)
    if (prior) {
        if (lds > 48 * 1024)
            FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(attn_fwd_k<true>, grid, dim3(256), lds, st, Q, K, v, in_lens, prior, attn, logprob, p_save, T, B, L, A, LP, 1.0f / temperature);
    } else {
        if (lds > 48 * 1024)
            FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(attn_fwd_k<false>, grid, dim3(256), lds, st, Q, K, v, in_lens, prior, attn, logprob, p_save, T, B, L, A, LP, 1.0f / temperature);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int ft_attention_bwd(const float* Q, const float* K, const float* v, const int32_t* in_lens, const float* prior,
                                const float* attn, const float* p_save, const float* dattn, const float* dlogprob,
                                float* de_work, float* dQ, float* dK, float* dv,
                                int T, int B, int L, int A, float temperature, void* stream) {
    FT_CHECK_ARG(Q && K && v && in_lens && attn && dattn && de_work && dQ && dK && dv);
    FT_CHECK_ARG(T >= 1 && B >= 1 && L >= 1 && A >= 1 && temperature > 0.f);
    FT_CHECK_ARG(prior == nullptr || p_save != nullptr);
    FT_CHECK_ARG(B <= 65535);
    const size_t lds_q = sizeof(float) * ((size_t)L * 32);
    if (lds_q > (size_t)MAX_LDS - 1024)
        return ft_fail(FT_EUNSUPPORTED, "ft_attention_bwd: L=%d exceeds the LDS tile (%zu B)", L, lds_q);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float inv_temp = 1.0f / temperature;
    const int rows_grid = cdiv((int64_t)B * T, 4);
    if (prior)
        hipLaunchKernelGGL(attn_softmax_bwd_k<true>, dim3(rows_grid), dim3(256), 0, st, attn, p_save, dattn, dlogprob, in_lens, de_work, T, B, L, inv_temp);
    else
        hipLaunchKernelGGL(attn_softmax_bwd_k<false>, dim3(rows_grid), dim3(256), 0, st, attn, p_save, dattn, dlogprob, in_lens, de_work, T, B, L, inv_temp);
    // one tanh pass for dQ, dK and dv (separate dQ / dK kernels, each recomputing tanh: 73.8 vs 71.5 ms per training step)
    FT_CHECK_HIP(hipMemsetAsync(dK, 0, sizeof(float) * (size_t)L * B * A, st));
    // four waves per workgroup = four 64-wide a-chunks.  Measured alternatives (profiles/r03_attn_dqdk_variants.log): five waves
    // (A = 640 = 2 x 5 chunks, no idle wave): 772 vs 603 us -- 134 VGPRs leave three waves per SIMD, i.e. three 4-wave
    // workgroups per CU but only two 5-wave ones; the dK sums of four tiles combined in LDS before leaving the CU (a quarter of
    // the atomics, 115 KB of LDS = one workgroup per CU): 1 045 us.  The pass is VALU-bound, not atomics-bound.
    const int nchunk = cdiv(A, 64);
    // ... and a workgroup of four waves whose last group covers fewer chunks carries idle waves (A = 640: 4 + 4 + 2): the largest of
    // 4 / 3 / 2 waves that divides the chunk count (round 6: A = 640 as five workgroups of two waves, 0.554 -> 0.521 ms per backward call)
    int nw = nchunk < 4 ? nchunk : 4;
    if (nchunk > 4) { if (nchunk % 4 == 0) nw = 4; else if (nchunk % 3 == 0) nw = 3; else if (nchunk % 2 == 0) nw = 2; }
    if (lds_q > 48 * 1024)      // (the kernel also has a few static LDS bytes: ask for what this launch needs, not for all 160 KiB)
        FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dqdk_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
    hipLaunchKernelGGL(attn_dqdk_k, dim3(cdiv(T, 32), B, cdiv(nchunk, nw)), dim3(64 * nw), lds_q, st, Q, K, v, in_lens, de_work, dQ, dK, dv,
                       T, B, L, A);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
