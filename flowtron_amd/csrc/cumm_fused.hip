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
constexpr int CKP = 128;                                   // row pitch of the col2 stream: 96 taps, a ONE (the bias gradient rides on the GEMM), zeros
constexpr int XW = 40;                                     // x staging: 38 positions (32 rows + 3 either side) per channel
constexpr int H1P = 33;                                    // h1 staging pitch (34 positions x 32 channels)
constexpr int FWD_ROWS = 32, BWD_OWN = 26, HALO = 3;

__device__ __forceinline__ bf16x8 ld_frag(const unsigned short* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack_op16x2(a, b), pack_op16x2(c, d)); }
// streamed-once traffic (the saved tanh, the 16-bit streams of the weight-gradient GEMMs) goes past the L2's working set -- the
// weight fragments, the lane-order text and its gradient, which every frame touches again -- with the non-temporal hint
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void nt_store16(void* p, const void* lds_src) {
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4_t*>(lds_src), reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ float4 nt_load16(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
// Workgroup barrier that publishes LDS writes only.  __syncthreads() also drains vmcnt: every frame kernel keeps weight fragments
// of LATER phases in flight across its barriers (requested at the kernel's top: a frame is a chain of short phases, each of which
// would otherwise start with an exposed L2 round trip), and nothing here communicates through global memory inside a launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// softmax of e[0 .. len) into ps (ps[l] = 0 for len <= l < Lp); red: 8 floats
// (e0 = e[tid], requested by the caller ahead of everything else)
__device__ __forceinline__ void softmax_block(const float* __restrict__ e, float e0, float* ps, float* red, int len, int Lp, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int l = tid; l < len; l += 256) { const float x = l == tid ? e0 : e[l]; ps[l] = x; m = fmaxf(m, x); }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    lds_barrier();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int l = tid; l < len; l += 256) { const float x = expf(ps[l] - m); ps[l] = x; s += x; }
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    lds_barrier();
    s = (red[4] + red[5]) + (red[6] + red[7]);
    for (int l = tid; l < Lp; l += 256) ps[l] = l < len ? ps[l] / s : 0.f;
    lds_barrier();
}

// h1s[jj][c], jj = 0 .. 33 <-> position l = r0 - 1 + jj: relu(b1[c] + sum_{ch,k} w1[c][ch][k] x[ch][l + k - 2]), 0 outside [0, L)
// (the second convolution zero-pads h1 at the ends).  xs[ch][pos] <-> position r0 - 3 + pos.  w = w1[c = tid & 31][:][:], bb = b1[c].
__device__ __forceinline__ void conv1_h1(const float* xs, float* h1s, const float (&w)[2 * K1], float bb, int r0, int L, int tid) {
    const int c = tid & 31;
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

// pre-activation of cond^T for one e-tile (weight fragments wa of its 3 k-steps) and the row tile behind cf: acc[r] <-> (e = 4 kg + r, l = li)
__device__ __forceinline__ f32x4 cond_pre(const bf16x8 (&wa)[3], const bf16x8 (&cf)[3]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = mfma16(wa[s], cf[s], acc);
    return acc;
}

// stage stamp k of frame i (workgroup (0, 0), thread 0): prof[(dir * 4096 + i) * 16 + k]
#define CUMMF_STAMP(dir, k) do { if (p.prof && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && i < 4096) p.prof[((dir) * 4096 + i) * 16 + (k)] = wall_clock64(); } while (0)

// Weight images are kept in MFMA FRAGMENT ORDER: the 16 x 32 block (row tile t, k-step s) of a [rows][K] matrix is one contiguous
// 1 KiB piece, lane (li, kg) -> 16 bytes at 16 * lane = W[16 t + li][32 s + 8 kg .. + 7]; piece index t * (K / 32) + s.  A fragment
// load is then ONE fully coalesced 1 KiB request.  (Read straight from the row-major image the same load touches 16 rows x 64
// bytes: the address unit takes it apart lane by lane -- 64 cycles instead of 16 -- and a frame was bound by exactly that:
// 17 us for the 20 k-steps of the key projection, profiles/r05_cumm_stage_stamps.log.)
constexpr int FRAG = 512;      // elements per fragment piece
__device__ __forceinline__ const unsigned short* frag_base(const unsigned short* img, int tile0, int ns, int lane) {
    return img + ((size_t)tile0 * ns * 64 + lane) * 8;
}

constexpr int PD = 4;          // k-steps of streamed weight fragments in flight per wave (16 bytes per lane and fragment)

// acc[q][rt] += W[rows 16 (wave + 4 q) + li][k] . tile[row 16 rt + li][k] over NS k-steps of 32: the weight fragments stream from the L2
// through a rotating window of PD k-steps (wf holds steps 0 .. PD-1 on entry: requested at the kernel's top), the activation tile
// sits in LDS (pitch KP elements).  Fully unrolled: every register index is static.
template <int NQ, int NS, int KP>
__device__ __forceinline__ void stream_gemm(f32x4 (&acc)[NQ][2], bf16x8 (&wf)[PD][NQ], const unsigned short* __restrict__ wp,
                                            const unsigned short* tile, int li, int kg) {
    const unsigned short* k0 = tile + li * KP + 8 * kg;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bf16x8 b0 = ld_frag(k0 + 32 * s), b1 = ld_frag(k0 + 16 * KP + 32 * s);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc[q][0] = mfma16(wf[s % PD][q], b0, acc[q][0]);
            acc[q][1] = mfma16(wf[s % PD][q], b1, acc[q][1]);
        }
        if (s + PD < NS) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) wf[s % PD][q] = ld_frag(wp + (size_t)q * 4 * NS * FRAG + FRAG * (s + PD));
        }
        // the machine scheduler otherwise sinks every request to right in front of its first use (register pressure), which turns the
        // window into one exposed L2 round trip per k-step
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
struct FwdP {
    const float *text, *Q, *v, *w1, *b1, *b2;
    const unsigned short *w2img, *wkimg;                  // [E][96], [A][E] 16-bit images
    const int* in_lens;
    float *attn, *logprob, *cumm_all, *tsave, *ebuf;      // ebuf [3][B][L]: scores of frame i accumulate in slot i % 3 (both column halves add)
    const float4* text_f;                                 // text in the lane order of this kernel's tiles (text_lane_k)
    const int* items;                                     // [0] = count, then one word per workgroup: (b << 16) | (j << 4) | (zh0 << 1) | (nz - 1)
    unsigned long long* gran;                             // persistent form: [2 parities][NSP][B][L] {epoch, partial score} granules
    int* status;                                          // persistent form: raised when a hand-off wait gives up
    int T, B, L, NJ;
    float inv_temp;
    long long* prof;                                      // debug (ft_cumm_debug_prof): stage stamps of workgroup (0, 0), 100 MHz clock
};

// One workgroup per ITEM = (utterance b, 32-row tile j, column halves [zh0, zh0 + nz)) of a list built once per call from in_lens
// (fwd_items_k): only tiles with valid rows, and -- when twice their number still fits the chip -- every tile as TWO workgroups that
// own one column half each (wave w: a-tiles w + 4 q of its half).  A frame is bound by what ONE CU can pull from the L2 (64 bytes
// per clock: 0.8 MB of W_key fragments per tile), and ~120 valid tiles leave half the chip idle; both halves redo the (small)
// location convolutions and add their partial scores.  (A plain 3-D grid dealt the valid tiles unevenly to the XCDs: one XCD with
// 33 of them on its 32 CUs made every frame two rounds.)
constexpr int NSP = 2;

// In-launch hand-off of the PERSISTENT forms (one launch walks many frames): the few hundred floats that cross workgroups per frame
// travel as 8-byte {epoch, value} granules -- the data is the flag (cdna_hip_programming.md G16 R2): one relaxed agent-scope store
// each (write-through: visible across XCDs), tag-checked relaxed agent-scope loads, no fences.  Two buffers alternate by the
// epoch's parity: a workgroup can run at most one frame ahead of the slowest workgroup of its utterance, so a buffer is rewritten
// only after everybody has read it.  Every spin is bounded (0.5 s of wall clock): a grid that is not co-resident raises the status word.
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
__device__ __forceinline__ void put_granule(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store((gu64_t*)g, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two values (the same slot of two granule arrays), waited for together; false = gave up (time-out, or somebody else's failure seen
// in *status)
__device__ __forceinline__ bool get_granules2(const unsigned long long* g0, const unsigned long long* g1, unsigned epoch, float& v0, float& v1,
                                              int* status, long long t0, long long limit) {
    for (int spin = 0;; ++spin) {
        const unsigned long long x0 = __hip_atomic_load((gu64_t*)g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long x1 = __hip_atomic_load((gu64_t*)g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch) {
            v0 = __uint_as_float((unsigned)x0); v1 = __uint_as_float((unsigned)x1);
            return true;
        }
        if ((spin & 63) == 63) {
            if (wall_clock64() - t0 > limit || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
constexpr long long PERSIST_TIMEOUT = 50000000LL;       // 0.5 s of the 100 MHz wall clock per wait

// PERSIST = false: frame i_first == i_last of one launch per frame (scores of frame i-1 from ebuf, running sum from cumm_all).
// PERSIST = true: ONE launch walks the frames i_first .. i_last (0 .. T): the work list guarantees a co-resident grid (<= one
// workgroup per CU), the partial scores of a tile's column halves cross workgroups as granules (gran [2][NSP][B][L]), the running sum
// lives in LDS, and what does not depend on the frame -- the w2 fragments, the text tile, v, b2, w1 -- is requested ONCE.
template <int NQE, int NQA, bool PERSIST>
__global__ __launch_bounds__(256, 1) void cummf_fwd_k(FwdP p, int i_first, int i_last) {
    constexpr int E = 64 * NQE, A = 64 * NQA, KPE = E + 8, NQH = NQA / NSP, AH = A / NSP;
    static_assert(NQA % NSP == 0 && AH == 64 * NQH, "column halves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= p.items[0]) return;
    const int item = p.items[1 + blockIdx.x];
    const int b = item >> 16, j = (item >> 4) & 0xfff, zh0 = (item >> 1) & 1, nz = (item & 1) + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int T = p.T, B = p.B, L = p.L, Lp = (L + 3) & ~3;
    const int len = min(p.in_lens[b], L);
    const int r0 = j * FWD_ROWS;
    const bool writer = j == 0 && zh0 == 0;               // closes frame i-1: attn, logprob, the running sum
    const bool has_rows = r0 < len;
    if (!has_rows && !writer) return;
    const size_t BL = (size_t)B * L;
    float* ps = reinterpret_cast<float*>(smem);           // [Lp]  attention of frame i-1
    float* xs = ps + Lp;                                  // [2][XW]
    float* h1s = xs + 2 * XW;                             // [34][H1P]
    float* red = h1s + 34 * H1P + 2;                      // [4][32] (first 8 also serve the softmax)
    float* qs = red + 128;                                // [A] Q_i[b]
    float* vs = qs + A;                                   // [A]
    float* b2s = vs + A;                                  // [E]
    unsigned short* kmt = reinterpret_cast<unsigned short*>(b2s + E);     // [32][KPE]   (float offset Lp + 1332 + 2 A + E: 16-byte aligned)
    float* tst = b2s + E + 32 * KPE / 2;                  // [32][AH + 4] fp32: the saved tanh of a column half on its way out
    float* cums = tst + 32 * (AH + 4);                    // [Lp] PERSIST: the running sum of the attention (cumm_i)
    const size_t RA = (size_t)L * B;

    // one launch per frame: the inputs of the softmax first (vmcnt retires in order: whatever is requested before them stands in front)
    float e_reg = -INFINITY, cpx = 0.f, cp0 = 0.f;        // score of frame i-1 at l = tid; cumm_{i-1} at this thread's x position / at l = tid
    if constexpr (!PERSIST) {
        if (i_first > 0) {
            const float* cprev = p.cumm_all + ((size_t)(i_first - 1) * B + b) * L;
            if (tid < len) e_reg = p.ebuf[(size_t)((i_first + 2) % 3) * BL + (size_t)b * L + tid];
            if (tid < 2 * 38) { const int l = r0 - HALO + (tid % 38); if (l >= 0 && l < L) cpx = cprev[l]; }
            if (writer && tid < L) cp0 = cprev[tid];
        }
    }
    // frame-independent requests: v / b2 / w1 (the w2 fragments and the text tile would stay too, but 200 more registers per lane
    // spill: they are requested again every frame)
    float w1r[2 * K1], b1r = 0.f;
    const unsigned short* w2p = frag_base(p.w2img, wave, 3, lane);
    const float4* txp = p.text_f + ((size_t)(b * p.NJ + j) * 4 + wave) * NQE * 2 * 64 + lane;
    if (has_rows) {
#pragma unroll
        for (int q = 0; q < 2 * K1; ++q) w1r[q] = p.w1[(tid & 31) * 2 * K1 + q];
        b1r = p.b1[tid & 31];
        if (tid < A / 4) *reinterpret_cast<float4*>(vs + 4 * tid) = *reinterpret_cast<const float4*>(p.v + 4 * tid);
        if (tid < E / 4) *reinterpret_cast<float4*>(b2s + 4 * tid) = *reinterpret_cast<const float4*>(p.b2 + 4 * tid);
    }
    if constexpr (PERSIST) { for (int l = tid; l < Lp; l += 256) cums[l] = 0.f; }
    long long t_wait = 0;

    const int i_end = PERSIST ? i_last : i_first;           // (one launch per frame: a single trip, known to the compiler)
#pragma unroll 1
    for (int i = i_first; i <= i_end; ++i) {
        const bool work = i < T && has_rows;
        float* e_out = p.ebuf + (size_t)(i % 3) * BL + (size_t)b * L;
        const float* cprev = i > 0 ? p.cumm_all + ((size_t)(i - 1) * B + b) * L : nullptr;
        CUMMF_STAMP(0, 0);
        // per-frame requests: the first PD k-steps of this wave's W_key rows, its w2 rows, Q_i -- they arrive under the softmax of the
        // previous frame
        bf16x8 wf[PD][NQH];
        bf16x8 w2f[NQE][3];
        float4 txv[NQE][2];
        float4 stage_q = make_float4(0.f, 0.f, 0.f, 0.f);
        // (the fragment addresses do not depend on the frame: without an opaque term the compiler hoists these requests out of the
        // frame loop of the persistent form and keeps 280 registers of weights alive across it -- 192 spilled)
        int opaque = 0;
        if constexpr (PERSIST) asm volatile("" : "+s"(opaque));
        const unsigned short* wkp = frag_base(p.wkimg, wave + 4 * NQH * zh0, E / 32, lane) + opaque;
        const unsigned short* w2q = w2p + opaque;
        if (work) {
#pragma unroll
            for (int d = 0; d < PD; ++d)
#pragma unroll
                for (int q = 0; q < NQH; ++q) wf[d][q] = ld_frag(wkp + (size_t)q * 4 * (E / 32) * FRAG + FRAG * d);
#pragma unroll
            for (int q = 0; q < NQE; ++q) {
#pragma unroll
                for (int s = 0; s < 3; ++s) w2f[q][s] = ld_frag(w2q + (size_t)q * 4 * 3 * FRAG + FRAG * s);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) txv[q][rt] = txp[(q * 2 + rt) * 64 + opaque];
            }
            if (tid < A / 4) stage_q = *reinterpret_cast<const float4*>(p.Q + ((size_t)i * B + b) * A + 4 * tid);
        }
        __builtin_amdgcn_sched_barrier(0);               // (requests stay up here)

        // 1. attention of the previous frame
        if (i > 0) {
            if constexpr (PERSIST) {
                // the partial scores of frame i-1 (epoch i) of every tile of this utterance, both column halves
                const unsigned long long* g = p.gran + (size_t)(i & 1) * NSP * BL + (size_t)b * L;
                if (t_wait == 0) t_wait = wall_clock64();
                bool ok = true;
                for (int l = tid; l < len; l += 256) {
                    float e0 = 0.f, e1 = 0.f;
                    ok = ok && get_granules2(g + l, g + BL + l, (unsigned)i, e0, e1, p.status, t_wait, PERSIST_TIMEOUT);
                    ps[l] = e0 + e1;
                }
                if (!__syncthreads_and(ok ? 1 : 0)) {     // somebody gave up: the launch cannot complete (grid not co-resident)
                    if (tid == 0) atomicMax(p.status, 3);
                    return;
                }
                t_wait = wall_clock64();
                softmax_block(ps, ps[tid < len ? tid : 0], ps, red, len, Lp, tid);
            } else {
                softmax_block(p.ebuf + (size_t)((i + 2) % 3) * BL + (size_t)b * L, e_reg, ps, red, len, Lp, tid);
            }
        } else {
            for (int l = tid; l < Lp; l += 256) ps[l] = 0.f;
            lds_barrier();
        }
        if constexpr (PERSIST) {
            if (i > 0) { for (int l = tid; l < Lp; l += 256) cums[l] += ps[l]; }        // cumm_i = cumm_{i-1} + attn_{i-1} (own elements: no barrier)
        } else {
            if (zh0 == 0 && has_rows && tid < FWD_ROWS && r0 + tid < L) p.ebuf[(size_t)((i + 1) % 3) * BL + (size_t)b * L + r0 + tid] = 0.f;
        }
        // the writer closes frame i-1 (one launch per frame: now; persistent: behind this frame's publication -- every workgroup of
        // the utterance waits for the writer's scores, nobody for these stores)
        auto close_prev = [&]() {
            const size_t row = ((size_t)b * T + (i - 1)) * L;
            for (int l = tid; l < L; l += 256) {
                const float pl = ps[l];
                p.attn[row + l] = pl;
                p.logprob[row + l] = logf(pl + 1e-8f);
                if (i < T) p.cumm_all[((size_t)i * B + b) * L + l] = PERSIST ? cums[l] : (l == tid ? cp0 : cprev[l]) + pl;
            }
        };
        if (writer && i > 0 && (!PERSIST || !work)) close_prev();
        if (!work) {
            if constexpr (PERSIST) { if (i < T) { lds_barrier(); continue; } }
            break;
        }
        CUMMF_STAMP(0, 1);
        if constexpr (PERSIST) lds_barrier();            // (cums complete for the x gather below)
        // 2. x = [cumm_i ; prev_i] for positions r0 - 3 .. r0 + 34, first convolution
        if (tid < 2 * 38) {
            const int ch = tid / 38, pos = tid - 38 * ch, l = r0 - HALO + pos;
            float x = 0.f;
            if (l >= 0 && l < L && i > 0) x = ch == 0 ? (PERSIST ? cums[l] : cpx + ps[l]) : ps[l];
            xs[ch * XW + pos] = x;
        }
        if (tid < A / 4) *reinterpret_cast<float4*>(qs + 4 * tid) = stage_q;
        lds_barrier();
        conv1_h1(xs, h1s, w1r, b1r, r0, L, tid);
        lds_barrier();
        CUMMF_STAMP(0, 2);
        // 3. cond = sigmoid(conv2(h1)), km = text . cond -> LDS tile [32 rows][E]
        {
            bf16x8 cf[2][3];
            col_frags(h1s, 0, li, kg, cf[0]);
            col_frags(h1s, 1, li, kg, cf[1]);
#pragma unroll
            for (int q = 0; q < NQE; ++q) {
                const int e0 = 16 * (wave + 4 * q) + 4 * kg;
                const float4 bv = *reinterpret_cast<const float4*>(b2s + e0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const f32x4 c = cond_pre(w2f[q], cf[rt]);
                    const float4 tx = txv[q][rt];
                    *reinterpret_cast<uint2*>(kmt + (16 * rt + li) * KPE + e0) =
                        pack4(tx.x * sigm(c[0] + bv.x), tx.y * sigm(c[1] + bv.y), tx.z * sigm(c[2] + bv.z), tx.w * sigm(c[3] + bv.w));
                }
            }
        }
        lds_barrier();
        CUMMF_STAMP(0, 3);
        for (int zz = 0; zz < nz; ++zz) {
            const int zh = zh0 + zz;
            if (zz > 0) {                                 // (unsplit items only: the other half's fragments, one exposed round trip)
                wkp = frag_base(p.wkimg, wave + 4 * NQH * zh, E / 32, lane) + opaque;
#pragma unroll
                for (int d = 0; d < PD; ++d)
#pragma unroll
                    for (int q = 0; q < NQH; ++q) wf[d][q] = ld_frag(wkp + (size_t)q * 4 * (E / 32) * FRAG + FRAG * d);
            }
            // 4. K^T = W_key km^T: wave w owns the a-tiles w, w + 4, ... of the column half; both row tiles
            f32x4 acc[NQH][2];
#pragma unroll
            for (int q = 0; q < NQH; ++q) { acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            stream_gemm<NQH, E / 32, KPE>(acc, wf, wkp, kmt, li, kg);
            if (zz == 0) CUMMF_STAMP(0, 4);
            // 5. t = tanh(Q_i + K) (saved), e = v . t / temperature
            float part[2] = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const int ah = 16 * (wave + 4 * q) + 4 * kg, a0 = AH * zh + ah;       // column inside this half / in A
                const float4 qv = *reinterpret_cast<const float4*>(qs + a0);
                const float4 vv = *reinterpret_cast<const float4*>(vs + a0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float4 tv;
                    tv.x = 1.f - 2.f * rsig(C2 * (qv.x + acc[q][rt][0]));
                    tv.y = 1.f - 2.f * rsig(C2 * (qv.y + acc[q][rt][1]));
                    tv.z = 1.f - 2.f * rsig(C2 * (qv.z + acc[q][rt][2]));
                    tv.w = 1.f - 2.f * rsig(C2 * (qv.w + acc[q][rt][3]));
                    part[rt] = fmaf(vv.x, tv.x, fmaf(vv.y, tv.y, fmaf(vv.z, tv.z, fmaf(vv.w, tv.w, part[rt]))));
                    *reinterpret_cast<float4*>(tst + (16 * rt + li) * (AH + 4) + ah) = tv;
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                part[rt] += __shfl_xor(part[rt], 16, 64);
                part[rt] += __shfl_xor(part[rt], 32, 64);
                if (kg == 0) red[wave * 32 + 16 * rt + li] = part[rt];
            }
            lds_barrier();
            if (tid < 32 && r0 + tid < len) {
                const float e = ((red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid])) * p.inv_temp;
                if constexpr (PERSIST) put_granule(p.gran + (size_t)((i + 1) & 1) * NSP * BL + (size_t)zh * BL + (size_t)b * L + r0 + tid, (unsigned)(i + 1), e);
                else atomicAdd(e_out + r0 + tid, e);       // (two addends: order-free)
            }
            // the saved tanh leaves row by row (a lane of the MFMA layout holds 16 bytes of 16 DIFFERENT rows: stored from there the
            // address unit takes every request apart)
            for (int idx = tid; idx < FWD_ROWS * (AH / 4); idx += 256) {
                const int row = idx / (AH / 4), c4 = idx - row * (AH / 4), l = r0 + row;
                if (l < len) nt_store16(p.tsave + ((size_t)i * RA + (size_t)l * B + b) * A + AH * zh + 4 * c4, tst + row * (AH + 4) + 4 * c4);
            }
            if (zz + 1 < nz || PERSIST) lds_barrier();    // (red / the tanh tile are written again)
        }
        if constexpr (PERSIST) { if (writer && i > 0) close_prev(); }
        CUMMF_STAMP(0, 5);
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------------------
struct BwdP {
    const float *text, *v, *w1, *b1, *b2;
    const unsigned short *w2img, *wkT, *w2T;              // [E][96], [E][A] (= W_key^T), [96][E] (= w2^T) 16-bit images
    const int* in_lens;
    const float *attn, *cumm_all, *tsave, *DV, *dattn, *dlogprob;
    float* gbuf;                                          // [2 parity][2: prev, cumm][B][L]
    float *dQ, *dv_part, *dw1_part, *db1_part;
    const float4* text_b;                                 // text in the lane order of this kernel's tiles (text_lane_k)
    float4* dtx;                                          // the gradient of text in the same order (own rows), accumulated over the frames
    unsigned short *dK_s, *km_s, *dp2_s, *col2_s;         // streams, [slot][rowbase[b] + l][A | E | E | 96]: VALID rows only, packed
    const int* rowbase;                                   // [B + 1] exclusive prefix of min(in_lens, L): rows of a frame = rowbase[B]
    int T, B, L;
    float inv_temp;
    long long* prof;
};

template <int NQE, int NQA>
__global__ __launch_bounds__(256, 1) void cummf_bwd_k(BwdP p, int i, int slot) {
    constexpr int E = 64 * NQE, A = 64 * NQA, KPE = E + 8, KPA = A + 8, DCP = CK + 1, NW1 = NF * 2 * K1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
    const int T = p.T, B = p.B, L = p.L, Lp = (L + 3) & ~3;
    const int len = min(p.in_lens[b], L);
    const int o0 = BWD_OWN * j, r0 = o0 - HALO;
    if (o0 > len) return;                                 // (row `len` itself still carries a gradient of the first convolution)
    const int wg = b * gridDim.x + j;
    float* ps = reinterpret_cast<float*>(smem);           // [Lp] attention of frame i
    float* ss = ps + Lp;                                  // [Lp] dp, then s
    float* gcs = ss + Lp;                                 // [Lp] carried gradient of cumm
    float* xs = gcs + Lp;                                 // [2][XW]
    float* h1s = xs + 2 * XW;                             // [34][H1P]
    float* red = h1s + 34 * H1P + 2;                      // [8]
    float* b2s = red + 8;                                 // [E]
    float* w1s = b2s + E;                                 // [NW1]
    float* dcs = w1s + NW1;                               // [4 waves][32][DCP]  dcol2 partials (K split over the waves)
    float* dp1 = dcs + 4 * 32 * DCP;                      // [32][H1P]  dpre1 (rows 1 .. 30)
    unsigned short* dkt = reinterpret_cast<unsigned short*>(dp1 + 32 * H1P);      // [32][KPA]
    unsigned short* dpt = dkt + 32 * KPA;                                         // [32][KPE]
    float* dqs = reinterpret_cast<float*>(dpt);           // [4 waves][dq A | dv A]: phase 2 only, before the dpre2 tile exists
    static_assert(8 * A * 4 <= 32 * KPE * 2, "the dq / dv partials must fit the dpre2 tile they borrow");
    const size_t RA = (size_t)L * B;
    const size_t fr = (size_t)slot * p.rowbase[B] + p.rowbase[b];      // stream row of (this frame, utterance b, l = 0)
    const float* g_in = p.gbuf + (size_t)((i + 1) & 1) * 2 * B * L;
    float* g_out = p.gbuf + (size_t)(i & 1) * 2 * B * L;

    CUMMF_STAMP(1, 0);
    // the inputs of the softmax backward first (vmcnt retires in order): attention of frame i, dctx . V, the carried gradients
    const size_t arow = ((size_t)b * T + i) * L;
    float in_p = 0.f, in_d = 0.f, in_gc = 0.f;
    if (tid < len) {
        in_p = p.attn[arow + tid];
        in_gc = g_in[(size_t)B * L + (size_t)b * L + tid];
        in_d = p.DV[arow + tid] + g_in[(size_t)b * L + tid] + in_gc;
        if (p.dattn) in_d += p.dattn[arow + tid];
        if (p.dlogprob) in_d += p.dlogprob[arow + tid] / (in_p + 1e-8f);
    }
    float in_x = 0.f;                                     // x_i at this thread's position (r0 - 3 + tid % 38)
    if (tid < 2 * 38) {
        const int ch = tid / 38, l = r0 - HALO + (tid - 38 * ch);
        if (l >= 0 && l < L) in_x = ch == 0 ? p.cumm_all[((size_t)i * B + b) * L + l] : (i > 0 ? p.attn[arow - L + l] : 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
    // 0. requests that do not depend on the carried gradients: the first PD k-steps of this wave's W_key^T rows, the saved tanh of
    //    the tile's rows, v, w1 / b1 / b2
    bf16x8 wf[PD][NQE];
    const unsigned short* wtp = frag_base(p.wkT, wave, A / 32, lane);
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int q = 0; q < NQE; ++q) wf[d][q] = ld_frag(wtp + (size_t)q * 4 * (A / 32) * FRAG + FRAG * d);
    const int ag = tid & 31, rg = tid >> 5;
    float4 tvr[4][A / 128], vv[A / 128];
#pragma unroll
    for (int m = 0; m < A / 128; ++m) vv[m] = *reinterpret_cast<const float4*>(p.v + 4 * ag + 128 * m);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int l = r0 + rg + 8 * n;
#pragma unroll
        for (int m = 0; m < A / 128; ++m) {
            tvr[n][m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l >= 0 && l < len) tvr[n][m] = nt_load16(p.tsave + ((size_t)i * RA + (size_t)l * B + b) * A + 4 * ag + 128 * m);
        }
    }
    float w1r[2 * K1];
#pragma unroll
    for (int q = 0; q < 2 * K1; ++q) w1r[q] = p.w1[(tid & 31) * 2 * K1 + q];
    const float b1r = p.b1[tid & 31];
    float4 stage_b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < E / 4) stage_b = *reinterpret_cast<const float4*>(p.b2 + 4 * tid);
    float stage_w1[2] = {0.f, 0.f};
    if (tid < NW1) stage_w1[0] = p.w1[tid];
    if (tid + 256 < NW1) stage_w1[1] = p.w1[tid + 256];
    __builtin_amdgcn_sched_barrier(0);                   // (requests stay up here)

    // 1. softmax backward: s_l = p_l (dp_l - sum_m p_m dp_m) / temperature
    float sum = 0.f;
    for (int l = tid; l < len; l += 256) {
        float pl = in_p, gc = in_gc, d = in_d;
        if (l != tid) {                                   // (L > 256 only)
            pl = p.attn[arow + l];
            gc = g_in[(size_t)B * L + (size_t)b * L + l];
            d = p.DV[arow + l] + g_in[(size_t)b * L + l] + gc;
            if (p.dattn) d += p.dattn[arow + l];
            if (p.dlogprob) d += p.dlogprob[arow + l] / (pl + 1e-8f);
        }
        ps[l] = pl; ss[l] = d; gcs[l] = gc;
        sum = fmaf(pl, d, sum);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    if (tid < E / 4) *reinterpret_cast<float4*>(b2s + 4 * tid) = stage_b;
    if (tid < NW1) w1s[tid] = stage_w1[0];
    if (tid + 256 < NW1) w1s[tid + 256] = stage_w1[1];
    // x_i for the positions r0 - 3 .. r0 + 34
    if (tid < 2 * 38) xs[(tid / 38) * XW + tid % 38] = in_x;
    lds_barrier();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int l = tid; l < len; l += 256) ss[l] = ps[l] * (ss[l] - sum) * p.inv_temp;
    conv1_h1(xs, h1s, w1r, b1r, r0, L, tid);
    lds_barrier();
    CUMMF_STAMP(1, 1);
    // col2 rows of the own positions -> stream (B operand of the dw2 GEMM)
    for (int idx = tid; idx < BWD_OWN * (CKP / 8); idx += 256) {      // 16 bytes (8 taps) per lane and store
        const int jo = idx / (CKP / 8), c8 = idx - (CKP / 8) * jo, l = o0 + jo;
        if (l < len) {
            float v[8];
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int ck = 8 * c8 + e8, c = ck / 3, k = ck - 3 * c;
                v[e8] = ck < CK ? h1s[(l - r0 + k) * H1P + c] : (ck == CK ? 1.f : 0.f);
            }
            const uint4 u = make_uint4(pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3]), pack_op16x2(v[4], v[5]), pack_op16x2(v[6], v[7]));
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, u), reinterpret_cast<u32x4_t*>(p.col2_s + (fr + l) * CKP + 8 * c8));
        }
    }
    CUMMF_STAMP(1, 8);
    // 2. dK = s v (1 - t^2) for the 32 computed rows -> LDS tile (B operand of the dkm GEMM) + stream (own rows);
    //    dQ_i = sum_l dK, dv += sum_l s t over the own rows
    {
        float4 dq[A / 128], dvp[A / 128];
#pragma unroll
        for (int m = 0; m < A / 128; ++m) { dq[m] = make_float4(0.f, 0.f, 0.f, 0.f); dvp[m] = dq[m]; }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = rg + 8 * n, l = r0 + row;
            const bool valid = l >= 0 && l < len;
            const float sl = valid ? ss[l] : 0.f;
            const bool own = valid && row >= HALO && row < HALO + BWD_OWN;
#pragma unroll
            for (int m = 0; m < A / 128; ++m) {
                const int a = 4 * ag + 128 * m;
                const float4 tv = tvr[n][m];
                float4 dk;
                dk.x = sl * vv[m].x * fmaf(-tv.x, tv.x, 1.f);
                dk.y = sl * vv[m].y * fmaf(-tv.y, tv.y, 1.f);
                dk.z = sl * vv[m].z * fmaf(-tv.z, tv.z, 1.f);
                dk.w = sl * vv[m].w * fmaf(-tv.w, tv.w, 1.f);
                const uint2 pk = pack4(dk.x, dk.y, dk.z, dk.w);
                *reinterpret_cast<uint2*>(dkt + row * KPA + a) = pk;
                if (own) {
                    *reinterpret_cast<uint2*>(p.dK_s + (fr + l) * A + a) = pk;
                    dq[m].x += dk.x; dq[m].y += dk.y; dq[m].z += dk.z; dq[m].w += dk.w;
                    dvp[m].x = fmaf(sl, tv.x, dvp[m].x); dvp[m].y = fmaf(sl, tv.y, dvp[m].y);
                    dvp[m].z = fmaf(sl, tv.z, dvp[m].z); dvp[m].w = fmaf(sl, tv.w, dvp[m].w);
                }
            }
        }
        // the two row groups of a wave combine by one shuffle, the four waves through LDS ([wave][2 A]: dq | dv)
#pragma unroll
        for (int m = 0; m < A / 128; ++m) {
            dq[m].x += __shfl_xor(dq[m].x, 32, 64); dq[m].y += __shfl_xor(dq[m].y, 32, 64);
            dq[m].z += __shfl_xor(dq[m].z, 32, 64); dq[m].w += __shfl_xor(dq[m].w, 32, 64);
            dvp[m].x += __shfl_xor(dvp[m].x, 32, 64); dvp[m].y += __shfl_xor(dvp[m].y, 32, 64);
            dvp[m].z += __shfl_xor(dvp[m].z, 32, 64); dvp[m].w += __shfl_xor(dvp[m].w, 32, 64);
            if (lane < 32) {
                *reinterpret_cast<float4*>(dqs + wave * 2 * A + 4 * ag + 128 * m) = dq[m];
                *reinterpret_cast<float4*>(dqs + wave * 2 * A + A + 4 * ag + 128 * m) = dvp[m];
            }
        }
    }
    lds_barrier();
    CUMMF_STAMP(1, 2);
    for (int a = tid; a < 2 * A; a += 256) {              // (fire and forget: dQ is shared by an utterance's tiles, the dv slot is this workgroup's own)
        const float x = (dqs[a] + dqs[2 * A + a]) + (dqs[4 * A + a] + dqs[6 * A + a]);
        if (a < A) atomicAdd(p.dQ + ((size_t)i * B + b) * A + a, x);
        else atomicAdd(p.dv_part + (size_t)wg * A + (a - A), x);
    }
    // 3. dkm^T = W_key^T dK^T: wave w owns the e-tiles w, w + 4, ...
    f32x4 acc[NQE][2];
#pragma unroll
    for (int q = 0; q < NQE; ++q) { acc[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    stream_gemm<NQE, A / 32, KPA>(acc, wf, wtp, dkt, li, kg);
    CUMMF_STAMP(1, 3);
    lds_barrier();                                        // everybody is done with the dK tile: the km tile goes over it
    unsigned short* kmt = dkt;                            // [32][KPE]
    // 4. cond again (MFMA, K = 96); km -> stream; dtext += dkm . cond; dpre2 = dkm . text . cond (1 - cond) -> LDS + stream; db2.
    //    All of the phase's reads are requested first; the w2^T fragments of phase 5 are requested as registers fall free.
    bf16x8 w2t[(E / 32 / 4) * (CK / 16)];                 // this wave's k-steps (wave, wave + 4, ...) x the 6 row tiles of w2^T
    {
        bf16x8 w2f[NQE][3];
        float4 txv[NQE][2], dold[NQE][2];
        const unsigned short* w2p = frag_base(p.w2img, wave, 3, lane);
        const size_t lo = ((size_t)(b * gridDim.x + j) * 4 + wave) * NQE * 2 * 64 + lane;       // this lane's slot in the lane-order images
#pragma unroll
        for (int q = 0; q < NQE; ++q) {
#pragma unroll
            for (int s = 0; s < 3; ++s) w2f[q][s] = ld_frag(w2p + (size_t)q * 4 * 3 * FRAG + FRAG * s);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = 16 * rt + li, l = r0 + row;
                txv[q][rt] = make_float4(0.f, 0.f, 0.f, 0.f); dold[q][rt] = txv[q][rt];
                if (l >= 0 && l < len) {                   // (beyond len: dK = 0, hence dkm = 0 and nothing of the row is kept)
                    txv[q][rt] = p.text_b[lo + (q * 2 + rt) * 64];
                    if (row >= HALO && row < HALO + BWD_OWN) dold[q][rt] = p.dtx[lo + (q * 2 + rt) * 64];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 cf[2][3];
        col_frags(h1s, 0, li, kg, cf[0]);
        col_frags(h1s, 1, li, kg, cf[1]);
        const unsigned short* w2tp = frag_base(p.w2T, 0, E / 32, lane) + (size_t)wave * FRAG;     // k-steps wave, wave + 4, ...
#pragma unroll
        for (int q = 0; q < NQE; ++q) {
            const int e0 = 16 * (wave + 4 * q) + 4 * kg;
            const float4 bv = *reinterpret_cast<const float4*>(b2s + e0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 cp = cond_pre(w2f[q], cf[rt]);
                const int row = 16 * rt + li, l = r0 + row;
                const bool inl = l >= 0 && l < len;                    // beyond len: dK = 0, hence dkm = 0
                const float4 tx = txv[q][rt];
                const float c0 = sigm(cp[0] + bv.x), c1 = sigm(cp[1] + bv.y), c2 = sigm(cp[2] + bv.z), c3 = sigm(cp[3] + bv.w);
                const float d0 = acc[q][rt][0], d1 = acc[q][rt][1], d2 = acc[q][rt][2], d3 = acc[q][rt][3];
                float4 dp;
                dp.x = d0 * tx.x * c0 * (1.f - c0); dp.y = d1 * tx.y * c1 * (1.f - c1);
                dp.z = d2 * tx.z * c2 * (1.f - c2); dp.w = d3 * tx.w * c3 * (1.f - c3);
                const uint2 pk = pack4(dp.x, dp.y, dp.z, dp.w);
                *reinterpret_cast<uint2*>(dpt + row * KPE + e0) = pk;
                *reinterpret_cast<uint2*>(kmt + row * KPE + e0) = pack4(tx.x * c0, tx.y * c1, tx.z * c2, tx.w * c3);
                if (inl && row >= HALO && row < HALO + BWD_OWN) {
                    float4 o = dold[q][rt];
                    o.x = fmaf(d0, c0, o.x); o.y = fmaf(d1, c1, o.y); o.z = fmaf(d2, c2, o.z); o.w = fmaf(d3, c3, o.w);
                    p.dtx[lo + (q * 2 + rt) * 64] = o;
                }
            }
            // w2^T fragments of the next phase: three per e-tile pass (k-step sk = q / 2 of this wave, row tiles 3 (q & 1) ..)
            if (q < 2 * (E / 32 / 4)) {
#pragma unroll
                for (int mm = 0; mm < 3; ++mm) {
                    const int sk = q / 2, m = 3 * (q & 1) + mm;
                    w2t[sk * (CK / 16) + m] = ld_frag(w2tp + (size_t)m * (E / 32) * FRAG + (size_t)4 * FRAG * sk);
                }
            }
        }
    }
    lds_barrier();
    CUMMF_STAMP(1, 4);
    // km and dpre2 of the own rows leave for the streams row by row, 16 bytes per lane (from the MFMA layout a request would cover
    // 8 bytes of 16 different rows each)
    for (int idx = tid; idx < BWD_OWN * (E / 8); idx += 256) {
        const int jo = idx / (E / 8), c8 = idx - jo * (E / 8), l = o0 + jo;
        if (l < len) {
            const size_t g = (fr + l) * E + 8 * c8;
            nt_store16(p.km_s + g, kmt + (HALO + jo) * KPE + 8 * c8);
            nt_store16(p.dp2_s + g, dpt + (HALO + jo) * KPE + 8 * c8);
        }
    }
    CUMMF_STAMP(1, 7);
    // 5. dcol2^T [96][32 rows] = w2^T dpre2^T, the K = E reduction split over the waves (partials side by side in LDS, summed by phase 6)
    {
        static_assert(NQE >= 2 * (E / 32 / 4), "the w2^T fragments are requested during the e-tile passes");
        f32x4 a3[CK / 16][2];
#pragma unroll
        for (int m = 0; m < CK / 16; ++m) { a3[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; a3[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const unsigned short* k0 = dpt + li * KPE + 8 * kg + 32 * wave;
#pragma unroll
        for (int sk = 0; sk < E / 32 / 4; ++sk) {
            const bf16x8 b0 = ld_frag(k0 + 128 * sk), b1 = ld_frag(k0 + 16 * KPE + 128 * sk);
#pragma unroll
            for (int m = 0; m < CK / 16; ++m) {
                a3[m][0] = mfma16(w2t[sk * (CK / 16) + m], b0, a3[m][0]);
                a3[m][1] = mfma16(w2t[sk * (CK / 16) + m], b1, a3[m][1]);
            }
        }
#pragma unroll
        for (int m = 0; m < CK / 16; ++m)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dcs[(wave * 32 + 16 * rt + li) * DCP + 16 * m + 4 * kg + r] = a3[m][rt][r];
    }
    lds_barrier();
    CUMMF_STAMP(1, 5);
    // 6. dh1[l][c] = sum_k dcol2[l - k + 1][c, k]; dpre1 = dh1 where h1 > 0 (rows 1 .. 30 of the tile)
    for (int idx = tid; idx < 30 * 32; idx += 256) {
        const int jj = 1 + (idx >> 5), c = idx & 31, l = r0 + jj;
        float d = 0.f;
        if (l >= 0 && l < L && h1s[(jj + 1) * H1P + c] > 0.f) {
#pragma unroll
            for (int w = 0; w < 4; ++w)
                d += dcs[(w * 32 + jj + 1) * DCP + 3 * c] + dcs[(w * 32 + jj) * DCP + 3 * c + 1] + dcs[(w * 32 + jj - 1) * DCP + 3 * c + 2];
        }
        dp1[jj * H1P + c] = d;
    }
    lds_barrier();
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
                for (int k = 0; k < K1; ++k) d = fmaf(w1s[c * 2 * K1 + ch * K1 + k], dp1[(jj - k + 2) * H1P + c], d);
            }
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        const int l = o0 + jo;
        if (o < 2 * BWD_OWN && part == 0 && l < len) {
            if (ch == 1) g_out[(size_t)b * L + l] = d;
            else g_out[(size_t)B * L + (size_t)b * L + l] = gcs[l] + d;
        }
    }
    // 8. dw1[c][ch][k] += sum_{own l} dpre1[l][c] x[ch][l + k - 2], db1[c] += sum_{own l} dpre1[l][c]
    for (int idx = tid; idx < NW1 + NF; idx += 256) {
        float d = 0.f;
        if (idx < NW1) {
            const int c = idx / (2 * K1), rest = idx - c * 2 * K1, ch = rest / K1, k = rest - ch * K1;
            for (int jj = HALO; jj < HALO + BWD_OWN; ++jj) d = fmaf(dp1[jj * H1P + c], xs[ch * XW + jj + k + 1], d);
            atomicAdd(p.dw1_part + (size_t)wg * NW1 + idx, d);
        } else {
            const int c = idx - NW1;
            for (int jj = HALO; jj < HALO + BWD_OWN; ++jj) d += dp1[jj * H1P + c];
            atomicAdd(p.db1_part + (size_t)wg * NF + c, d);
        }
    }
    CUMMF_STAMP(1, 6);
}

// work list of the forward frames (one thread: B is small): tiles with valid rows only; every tile as two column-half workgroups
// when 2 x tiles <= n_cu, else as one workgroup that walks both halves
__global__ void fwd_items_k(const int* __restrict__ in_lens, int B, int L, int n_cu, int* __restrict__ items) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int nt = 0;
    for (int b = 0; b < B; ++b) { const int len = min(max(in_lens[b], 1), L); nt += (len + FWD_ROWS - 1) / FWD_ROWS; }
    const bool split = NSP * nt <= n_cu;
    int n = 0;
    for (int b = 0; b < B; ++b) {
        const int len = min(max(in_lens[b], 1), L), tiles = (len + FWD_ROWS - 1) / FWD_ROWS;
        for (int j = 0; j < tiles; ++j) {
            if (split) { for (int z = 0; z < NSP; ++z) items[1 + n++] = (b << 16) | (j << 4) | (z << 1); }
            else items[1 + n++] = (b << 16) | (j << 4) | (NSP - 1);
        }
    }
    items[0] = n;
}

// fragment-order 16-bit image of the logical matrix W [rows][K] (rows % 16 == 0, K % 32 == 0): W = src (row-major [rows][K]) or, with
// `transpose`, W[r][c] = src[c][r] (src row-major [K][rows])
__global__ void cvt16_frag_k(const float* __restrict__ src, int rows, int K, unsigned short* __restrict__ dst, int transpose) {
    const long n = (long)rows * K;
    const int ns = K / 32;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int r = (int)(q / K), c = (int)(q - (long)r * K);
        const float v = transpose ? src[(size_t)c * rows + r] : src[q];
        dst[((size_t)((r >> 4) * ns + (c >> 5)) * 64 + ((c & 31) >> 3) * 16 + (r & 15)) * 8 + (c & 7)] = f2op16(v);
    }
}
// text [L][B][E] in the lane order of the frame kernels: out[(((b NJ + j) 4 + w) NQ + q) 2 + rt][lane] = the float4
// text[l = own * j + off + 16 rt + li][b][16 (w + 4 q) + 4 kg ..], zeros outside [0, L)   (NQ = E / 64)
__global__ void text_lane_k(const float* __restrict__ text, int B, int L, int E, int NJ, int own, int off, float4* __restrict__ out) {
    const int NQ = E / 64;
    const long n = (long)B * NJ * 4 * NQ * 2 * 64;
    for (long x = blockIdx.x * (long)blockDim.x + threadIdx.x; x < n; x += (long)gridDim.x * blockDim.x) {
        long y = x;
        const int lane = (int)(y & 63); y >>= 6;
        const int rt = (int)(y & 1); y >>= 1;
        const int q = (int)(y % NQ); y /= NQ;
        const int w = (int)(y & 3); y >>= 2;
        const int j = (int)(y % NJ), b = (int)(y / NJ);
        const int l = own * j + off + 16 * rt + (lane & 15), e0 = 16 * (w + 4 * q) + 4 * (lane >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l >= 0 && l < L) v = *reinterpret_cast<const float4*>(text + ((size_t)l * B + b) * E + e0);
        out[x] = v;
    }
}
// the inverse for the gradient: dtext[l][b][e0 ..] = the float4 of l's OWNER tile (j = l / own), zero beyond in_lens[b]
__global__ void dtext_gather_k(const float4* __restrict__ dtx, const int* __restrict__ in_lens, int B, int L, int E, int NJ, float* __restrict__ dtext) {
    const int NQ = E / 64, E4 = E / 4;
    const long n = (long)L * B * E4;
    for (long x = blockIdx.x * (long)blockDim.x + threadIdx.x; x < n; x += (long)gridDim.x * blockDim.x) {
        const int e4 = (int)(x % E4);
        const long lb = x / E4;
        const int b = (int)(lb % B), l = (int)(lb / B);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < in_lens[b]) {
            const int j = l / BWD_OWN, row = l - BWD_OWN * j + HALO, rt = row >> 4, li = row & 15;
            const int t = e4 >> 2, kg = e4 & 3, w = t & 3, q = t >> 2;
            v = dtx[((((size_t)(b * NJ + j) * 4 + w) * NQ + q) * 2 + rt) * 64 + kg * 16 + li];
        }
        *reinterpret_cast<float4*>(dtext + ((size_t)l * B + b) * E + 4 * e4) = v;
    }
}
// out[c] = sum_w part[w][c]
__global__ __launch_bounds__(256) void part_sum_k(const float* __restrict__ part, int nw, int n, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), y = threadIdx.x >> 6;
    float s = 0.f;
    if (c < n) for (int w = y; w < nw; w += 4) s += part[(size_t)w * n + c];
    red[y][threadIdx.x & 63] = s;
    __syncthreads();
    if (y == 0 && c < n) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// dw2 [E][CK] and db2 [E] out of the GEMM's [E][CKP] result (column CK = the ones column's product = column sums of dpre2)
__global__ void dw2_split_k(const float* __restrict__ x, int E, float* __restrict__ dw2, float* __restrict__ db2) {
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= (long)E * (CK + 1)) return;
    const int e = (int)(q / (CK + 1)), ck = (int)(q - (long)e * (CK + 1));
    if (ck < CK) dw2[(size_t)e * CK + ck] = x[(size_t)e * CKP + ck];
    else db2[e] = x[(size_t)e * CKP + CK];
}
// per call: rowbase [B + 1] (packed stream rows), rows_dev = {rows of a full chunk, rows of the top chunk}, and zeros in the 64 rows
// behind either extent of the four streams (the GEMMs' last 32-row k-step and their tile over-reads end there; every row in front of
// an extent is written by its owner tile in every chunk)
__global__ __launch_bounds__(256) void bwd_setup_k(const int* __restrict__ in_lens, int B, int L, int Tc, int nf_top, int* __restrict__ rowbase,
                                                   int* __restrict__ rows_dev, unsigned short* dK_s, unsigned short* km_s, unsigned short* dp2_s,
                                                   unsigned short* col2_s, int A, int E) {
    __shared__ int rv;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < B; ++b) { rowbase[b] = acc; acc += min(max(in_lens[b], 0), L); }
        rowbase[B] = acc;
        rows_dev[0] = Tc * acc; rows_dev[1] = nf_top * acc;
        rv = acc;
    }
    __syncthreads();
    for (int e = 0; e < 2; ++e) {
        const size_t r0 = (size_t)(e == 0 ? Tc : nf_top) * rv;
        for (size_t x = threadIdx.x; x < (size_t)64 * A; x += 256) dK_s[r0 * A + x] = 0;
        for (size_t x = threadIdx.x; x < (size_t)64 * E; x += 256) { km_s[r0 * E + x] = 0; dp2_s[r0 * E + x] = 0; }
        for (size_t x = threadIdx.x; x < (size_t)64 * CKP; x += 256) col2_s[r0 * CKP + x] = 0;
    }
}

inline size_t up256(size_t v) { return (v + 255) & ~size_t(255); }

struct Carve {
    unsigned short *w2img, *wkimg, *wkT, *w2T;
    float *ebuf, *gbuf, *DV, *dv_part, *dw1_part, *db1_part;
    int* items;                                            // forward work list
    unsigned long long* gran;                              // forward, persistent form: score granules
    int *rowbase, *rows_dev;                               // backward: packed stream rows
    float4 *text_l, *dtx;                                  // lane-order text (forward or backward tiling) and its gradient
    size_t lane_bytes;
    unsigned short *dK_s, *km_s, *dp2_s, *col2_s;
    float* dw2x;                                           // [E][CKP]: dw2 | db2 (column CK) as the GEMM leaves them
    int Tc, nwg;
    size_t part_floats, stream_bytes, total;
};

Carve carve(void* base, int T, int L, int B, int E, int A, bool bwd) {
    Carve c{};
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = base ? reinterpret_cast<char*>(base) + off : nullptr; off += up256(bytes); return q; };
    c.w2img = reinterpret_cast<unsigned short*>(take((size_t)E * CK * 2 + 256));
    c.wkimg = reinterpret_cast<unsigned short*>(take((size_t)A * E * 2 + 256));
    c.ebuf = reinterpret_cast<float*>(take((size_t)3 * B * L * 4));
    c.items = reinterpret_cast<int*>(take(sizeof(int) * (1 + (size_t)NSP * B * cdiv(L, FWD_ROWS))));
    c.gran = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)2 * NSP * B * L));
    c.lane_bytes = (size_t)B * cdiv(L, bwd ? BWD_OWN : FWD_ROWS) * 4 * (E / 64) * 2 * 64 * sizeof(float4);
    c.text_l = reinterpret_cast<float4*>(take(c.lane_bytes));
    if (bwd) {
        c.dtx = reinterpret_cast<float4*>(take(c.lane_bytes));
        c.rowbase = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)B + 1)));
        c.rows_dev = reinterpret_cast<int*>(take(sizeof(int) * 2));
        c.wkT = reinterpret_cast<unsigned short*>(take((size_t)A * E * 2 + 256));
        c.w2T = reinterpret_cast<unsigned short*>(take((size_t)E * CK * 2 + 256));
        c.gbuf = reinterpret_cast<float*>(take((size_t)4 * B * L * 4));
        c.DV = reinterpret_cast<float*>(take((size_t)B * T * L * 4));
        c.nwg = B * cdiv(L, BWD_OWN);
        c.part_floats = (size_t)c.nwg * (A + NF * 2 * K1 + NF);
        c.dv_part = reinterpret_cast<float*>(take(c.part_floats * 4));
        c.dw1_part = c.dv_part ? c.dv_part + (size_t)c.nwg * A : nullptr;
        c.db1_part = c.dv_part ? c.dw1_part + (size_t)c.nwg * NF * 2 * K1 : nullptr;
        // frames per chunk: <= ~1.6 GB of streams at full-length utterances (the weight-gradient GEMMs run once per chunk)
        const size_t per_frame = (size_t)L * B * (A + 2 * E + CKP) * 2;
        long tc = (long)(1600000000ull / per_frame);
        const char* tc_env = getenv("FT_CUMM_CHUNK");          // test hook: frames per chunk (the multi-chunk / two-set path at small T)
        if (tc_env && atoi(tc_env) > 0) tc = atoi(tc_env);
        c.Tc = (int)(tc < 1 ? 1 : (tc > T ? T : tc));
        const size_t rows = (size_t)c.Tc * L * B + 320;             // slack: the GEMM tiles read up to 256 columns / 32 rows past the end
        const size_t s0 = off;
        c.dK_s = reinterpret_cast<unsigned short*>(take(rows * A * 2));
        c.km_s = reinterpret_cast<unsigned short*>(take(rows * E * 2));
        c.dp2_s = reinterpret_cast<unsigned short*>(take(rows * E * 2));
        c.col2_s = reinterpret_cast<unsigned short*>(take(rows * CKP * 2));
        c.stream_bytes = off - s0;
        c.dw2x = reinterpret_cast<float*>(take((size_t)E * CKP * 4));
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

long long* g_cummf_prof = nullptr;

#define CK_(x) do { int rc_ = (x); if (rc_ != FT_OK) return rc_; } while (0)

}  // namespace

// debug hook: device buffer [2][4096][16] int64 that later fused launches of THIS operand format fill with the stage stamps of
// workgroup (0, 0) (100 MHz wall clock); NULL switches it off
void FT_OPNAME(ftint_cummf_debug_prof)(void* dev_buf) { g_cummf_prof = reinterpret_cast<long long*>(dev_buf); }

// shapes the fused kernels are instantiated for (config.json: n_text_dim 512 + n_speaker_dim 128 = 640 = n_attn_channels)
int FT_OPNAME(ftint_cummf_supported)(const ft_cumm_attn_args* a) {
    return a->mode == FT_OP16 && a->E == 640 && a->A == 640 && a->NF == NF && a->K1 == K1 && a->K2 == K2 && a->L <= 1024 && a->B < 32768;    // (L: the backward tile needs 12 L + 146 KB of LDS)
}

size_t FT_OPNAME(ftint_cummf_workspace_bytes)(int T, int L, int B, int E, int A, int backward) {
    return carve(nullptr, T, L, B, E, A, backward != 0).total;
}

int FT_OPNAME(ftint_cummf_fwd)(const ft_cumm_attn_args* a, hipStream_t st) {
    const int T = a->T, B = a->B, L = a->L, E = a->E, A = a->A;
    const Carve c = carve(a->work, T, L, B, E, A, false);
    FT_CHECK_ARG(a->work_bytes >= c.total);
    hipLaunchKernelGGL(cvt16_frag_k, dim3(240), dim3(256), 0, st, a->w2, E, CK, c.w2img, 0);
    hipLaunchKernelGGL(cvt16_frag_k, dim3(1024), dim3(256), 0, st, a->w_key, A, E, c.wkimg, 0);
    hipLaunchKernelGGL(text_lane_k, dim3(2048), dim3(256), 0, st, a->text, B, L, E, cdiv(L, FWD_ROWS), FWD_ROWS, 0, c.text_l);
    FT_CHECK_HIP(hipMemsetAsync(a->cumm_all, 0, sizeof(float) * (size_t)B * L, st));           // cumm_0 = 0
    FT_CHECK_HIP(hipMemsetAsync(c.ebuf, 0, sizeof(float) * (size_t)3 * B * L, st));
    static int n_cu = -1;
    if (n_cu < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const char* split_env = getenv("FT_CUMM_SPLIT");       // 0: never split a tile's columns over two workgroups (tests both item forms)
    hipLaunchKernelGGL(fwd_items_k, dim3(1), dim3(64), 0, st, a->in_lens, B, L, (split_env && atoi(split_env) == 0) ? 0 : n_cu, c.items);
    FwdP p{};
    p.text = a->text; p.Q = a->Q; p.v = a->v; p.w1 = a->w1; p.b1 = a->b1; p.b2 = a->b2; p.w2img = c.w2img; p.wkimg = c.wkimg;
    p.in_lens = a->in_lens; p.attn = a->attn; p.logprob = a->logprob; p.cumm_all = a->cumm_all; p.tsave = a->kproj_all; p.ebuf = c.ebuf; p.text_f = c.text_l; p.items = c.items;
    p.T = T; p.B = B; p.L = L; p.NJ = cdiv(L, FWD_ROWS); p.inv_temp = 1.0f / a->temperature; p.prof = g_cummf_prof;
    const int Lp = (L + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)2 * Lp + 2 * XW + 34 * H1P + 2 + 128 + 2 * A + E) + (size_t)32 * (E + 8) * 2 + sizeof(float) * 32 * (A / NSP + 4);
    FT_CHECK_ARG(B < 32768 && cdiv(L, FWD_ROWS) < 4096);
    const dim3 grid(NSP * B * cdiv(L, FWD_ROWS));
    // ONE persistent launch for all frames where the work list is co-resident by construction: it holds min(2 x valid tiles, n_cu)
    // or (unsplit) valid tiles <= B x ceil(L / 32) workgroups of one per CU (96 KB of LDS), the rest of the grid exits at once.
    // Needs the caller's status word (a->persist_status: the wait that gives up raises it; FT_CUMM_PERSIST=0 keeps one launch per frame).
    const char* pe = getenv("FT_CUMM_PERSIST");
    const bool persist = a->persist_status != nullptr && !(pe && atoi(pe) == 0) && B * cdiv(L, FWD_ROWS) <= n_cu;
    if (persist) {
        FT_CHECK_HIP(hipMemsetAsync(c.gran, 0, sizeof(unsigned long long) * (size_t)2 * NSP * B * L, st));       // epoch 0: never awaited
        p.gran = c.gran; p.status = a->persist_status;
        FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cummf_fwd_k<10, 10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((cummf_fwd_k<10, 10, true>), grid, dim3(256), lds, st, p, 0, T);
    } else {
        FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cummf_fwd_k<10, 10, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int i = 0; i <= T; ++i)                 // launch T only closes frame T-1 (softmax, attn, logprob)
            hipLaunchKernelGGL((cummf_fwd_k<10, 10, false>), grid, dim3(256), lds, st, p, i, i);
    }
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
    hipLaunchKernelGGL(cvt16_frag_k, dim3(240), dim3(256), 0, st, a->w2, E, CK, c.w2img, 0);
    hipLaunchKernelGGL(cvt16_frag_k, dim3(240), dim3(256), 0, st, a->w2, CK, E, c.w2T, 1);           // w2^T [96][E]
    hipLaunchKernelGGL(cvt16_frag_k, dim3(1024), dim3(256), 0, st, a->w_key, E, A, c.wkT, 1);        // W_key^T [E][A]
    hipLaunchKernelGGL(text_lane_k, dim3(2048), dim3(256), 0, st, a->text, B, L, E, cdiv(L, BWD_OWN), BWD_OWN, -HALO, c.text_l);
    FT_CHECK_HIP(hipMemsetAsync(c.dtx, 0, c.lane_bytes, st));
    FT_CHECK_HIP(hipMemsetAsync(c.gbuf, 0, sizeof(float) * (size_t)4 * B * L, st));
    FT_CHECK_HIP(hipMemsetAsync(c.dv_part, 0, sizeof(float) * c.part_floats, st));
    {
        const int n_chunks = cdiv(T, c.Tc), nf_top = T - (n_chunks - 1) * c.Tc;
        hipLaunchKernelGGL(bwd_setup_k, dim3(1), dim3(256), 0, st, a->in_lens, B, L, c.Tc, nf_top, c.rowbase, c.rows_dev, c.dK_s, c.km_s, c.dp2_s,
                           c.col2_s, A, E);
    }
    FT_CHECK_HIP(hipMemsetAsync(dQ, 0, sizeof(float) * (size_t)T * B * A, st));
    // DV[b][t][l] = dctx[t][b] . V[l][b]  and  dV[l][b][:] = sum_t attn[b][t][l] dctx[t][b][:]
    CK_(bgemm(dctx, a->V, c.DV, T, L, A, (long)B * A, 1, 1, (long)B * A, L, B, A, A, (long)T * L, a->mode, st));
    CK_(bgemm(a->attn, dctx, dV, L, A, T, 1, L, (long)B * A, 1, (long)B * A, B, (long)T * L, A, A, a->mode, st));
    BwdP p{};
    p.text = a->text; p.v = a->v; p.w1 = a->w1; p.b1 = a->b1; p.b2 = a->b2; p.w2img = c.w2img; p.wkT = c.wkT; p.w2T = c.w2T;
    p.in_lens = a->in_lens; p.attn = a->attn; p.cumm_all = a->cumm_all; p.tsave = a->kproj_all; p.DV = c.DV; p.dattn = dattn; p.dlogprob = dlogprob;
    p.gbuf = c.gbuf; p.dQ = dQ; p.text_b = c.text_l; p.dtx = c.dtx; p.dv_part = c.dv_part; p.dw1_part = c.dw1_part; p.db1_part = c.db1_part;
    p.dK_s = c.dK_s; p.km_s = c.km_s; p.dp2_s = c.dp2_s; p.col2_s = c.col2_s; p.rowbase = c.rowbase;
    p.T = T; p.B = B; p.L = L; p.inv_temp = 1.0f / a->temperature; p.prof = g_cummf_prof;
    const int Lp = (L + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)3 * Lp + 2 * XW + 34 * H1P + 2 + 8 + E + NF * 2 * K1 + 4 * 32 * (CK + 1) + 32 * H1P) +
                       (size_t)32 * (A + 8) * 2 + (size_t)32 * (E + 8) * 2;
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cummf_bwd_k<10, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(cdiv(L, BWD_OWN), B);
    const size_t RA = (size_t)L * B;
    // (Measured and NOT taken: the chunk GEMMs on a side stream confined to 4 .. 24 CUs per XCD (hipExtStreamCreateWithCUMask), beside
    // the next chunk's frames, which occupy only ~150 CUs: 155 - 190 ms per training step against 125 with everything in stream
    // order -- the frames are bound by what their CU pulls from the L2, and the GEMMs' traffic sits in front of it.)
    int k = 0;
    for (int hi = T; hi > 0; ++k) {
        const int lo = ((hi - 1) / c.Tc) * c.Tc, nf = hi - lo;
        for (int i = hi - 1; i >= lo; --i) hipLaunchKernelGGL((cummf_bwd_k<10, 10>), grid, dim3(256), lds, st, p, i, i - lo);
        FT_CHECK_LAUNCH();
        // weight gradients of the chunk: both operands k-major (the reduction runs over frames x VALID rows: K is the capacity,
        // k-steps beyond the device-side row count are not visited -- ft_gemm_img's compact reduction), split-K.  The col2 stream
        // carries a column of ones behind its 96 taps, so the second GEMM leaves db2 = column sums of dpre2 in column CK.
        const long Kr = ((long)nf * (long)RA + 31) / 32 * 32;
        ft_gemm_img_args g{};
        g.alpha = 1.f; g.beta = k == 0 ? 0.f : 1.f; g.act = FT_ACT_NONE; g.flags = FT_GEMM_SPLITK; g.a_kmajor = 1; g.b_kmajor = 1;
        g.K = (int)Kr; g.compact = 2; g.k_shift = 0; g.rows_dev = c.rows_dev + (nf == c.Tc ? 0 : 1);
        g.A = c.dK_s; g.lda = A; g.B = c.km_s; g.ldb = E; g.C = dw_key; g.ldc = E; g.M = A; g.N = E;
        CK_(FT_OPNAME(ft_gemm_img)(&g, st));
        g.A = c.dp2_s; g.lda = E; g.B = c.col2_s; g.ldb = CKP; g.C = c.dw2x; g.ldc = CKP; g.M = E; g.N = CK + 1;
        CK_(FT_OPNAME(ft_gemm_img)(&g, st));
        hi = lo;
    }
    hipLaunchKernelGGL(dw2_split_k, dim3(cdiv((long)E * (CK + 1), 256)), dim3(256), 0, st, c.dw2x, E, dw2, db2);
    hipLaunchKernelGGL(dtext_gather_k, dim3(2048), dim3(256), 0, st, c.dtx, a->in_lens, B, L, E, cdiv(L, BWD_OWN), dtext);
    hipLaunchKernelGGL(part_sum_k, dim3(cdiv(A, 64)), dim3(256), 0, st, c.dv_part, c.nwg, A, dv);
    hipLaunchKernelGGL(part_sum_k, dim3(cdiv(NF * 2 * K1, 64)), dim3(256), 0, st, c.dw1_part, c.nwg, NF * 2 * K1, dw1);
    hipLaunchKernelGGL(part_sum_k, dim3(1), dim3(256), 0, st, c.db1_part, c.nwg, NF, db1);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
