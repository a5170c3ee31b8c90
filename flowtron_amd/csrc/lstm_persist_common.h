// Helpers shared by the persistent recurrence kernels (lstm_persist.hip: the round-2..5 kernels, 4 batch rows per XCD group;
// lstm_roles.hip: round 6 -- R = 4 / 8 / 16 rows per group, time windows with carried state, two roles per launch).
#pragma once
#include "common.h"
#include "lstm_images.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) void gl_void;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int PH = 1024;                 // hidden size this kernel is built for
constexpr int NCHUNK = PH / 32;          // 32 k-chunks of 32
constexpr int NCU = 256;

// One LDS-DMA dword per lane (global_load_lds_dword: LDS address = M0 base + 4 lane, no VGPR staging) as inline asm: hipcc
// put an s_waitcnt vmcnt(0) in front of every builtin DMA of a burst (64 serial HBM round trips); an asm statement is not
// counted, so the caller waits once, `s_waitcnt vmcnt(0)`, after the last one.  M0 is saved and restored in the statement
// (cdna_hip_programming.md 5.7).  lds_addr must be wave-uniform.
__device__ __forceinline__ void dma_dword(const float* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}

// lane l <- lane l + n of the same 16-lane row (DPP row_shl:n), n = 0 .. 15
template <int N>
__device__ __forceinline__ unsigned row_shl(unsigned v) {
    if constexpr (N == 0) return v;
    else return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}

// BARE hand-off (template BARE = true): the same packed layout WITHOUT tags -- a 16-byte load carries eight k-values, half the
// bytes per step (measured: a wave reading freshly written granules is latency-bound at ~50-75 GB/s per CU, so the 64 KB a CU
// pulls per backward step are most of its 2 us sweep).  "Not yet written" is a SENTINEL instead of an epoch tag: three rotating
// buffers; during step n a producer resets its own slots of buffer (n+1) % 3 to 0xFFFFFFFF (that buffer holds step n-2, which
// every consumer of the group has finished reading: they all published step n-1, and this CU has gathered all of it), waits for
// those stores (s_waitcnt vmcnt(0)) and only then publishes step n into buffer n % 3.  A consumer that polls buffer (n+1) % 3
// for step n+1 has already gathered step n from every producer, hence stands behind every producer's reset: it can only see
// the sentinel or the new value, never the value of step n-2.  A dword is two 16-bit operands; a pair of NaNs with all-ones
// payloads (the converters produce the canonical 0x7FC0 / 0xFFC0 / 0x7E00) would read as "not yet" and end in the bounded
// time-out like any other failure.  Buffer layout [group][buffers][w][lg][64 lanes][4 dwords].  (Round 6, lstm_roles.hip: FOUR rotating
// buffers -- the reset of a step targets the buffer two steps ahead and is issued BEHIND the publish, so the wait in front of a
// publish covers a reset that is a whole step old instead of one just issued.)
constexpr unsigned SENT = 0xFFFFFFFFu;
template <int RPGP, int NCW>
__device__ __forceinline__ int bare_index(int b, int k) {       // dword index of the operand pair (k, k+1), k even
    constexpr int CPL = 16 / RPGP, NLG = NCW / CPL;
    const int c = k >> 5, w = c & 3, ci = c >> 2, kg = (k >> 3) & 3, e = k & 7;
    const int lg = ci / CPL, j = ci % CPL, lane = kg * 16 + j * RPGP + b;
    return (((w * NLG + lg) * 64 + lane) << 2) + (e >> 1);
}
template <int S>
__device__ __forceinline__ u32x4 shl4(u32x4 v) {
    return (u32x4){row_shl<S>(v[0]), row_shl<S>(v[1]), row_shl<S>(v[2]), row_shl<S>(v[3])};
}
// member j (0 .. CPL-1) of a load group moved into the MFMA row positions: a row shift by j RPGP lanes
template <int RPGP>
__device__ __forceinline__ u32x4 member(u32x4 v, int j) {
    switch (j) {
        case 0: return v;
        case 1: return shl4<RPGP & 15>(v);
        case 2: return shl4<(2 * RPGP) & 15>(v);
        default: return shl4<(3 * RPGP) & 15>(v);
    }
}

// XCD census (LOCAL transport): group = this workgroup's XCC id, slot = arrival order inside that XCD
template <int CPG>
__device__ __forceinline__ bool join_group_local(unsigned* census, int* status, int tid, int& grp, int& q) {
    __shared__ int slot[2];
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        slot[0] = (int)(xcc & 7u);
        slot[1] = (int)__hip_atomic_fetch_add(census + (xcc & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    grp = __builtin_amdgcn_readfirstlane(slot[0]);
    q = __builtin_amdgcn_readfirstlane(slot[1]);
    if (q >= CPG) {                                              // more than CPG workgroups on one XCD: not the machine this is for
        if (tid == 0) atomicExch(status, 2);
        return false;
    }
    return true;
}

// 16x16x32 MFMA with the B operand in ACCUMULATION registers and the accumulator in architectural ones (inline asm: the builtin
// only takes B from VGPRs, so fragments parked in AGPRs cost four v_accvgpr_read each per use).  lstm_persist_bwd_rs_k keeps all 64
// weight fragments of a wave (256 registers) in AGPRs for the whole launch and everything else in VGPRs: no register-file moves in
// the step.  The compiler does not see an MFMA here: the CALLER keeps dependent uses of `acc` far enough apart (>= 3 other MFMAs
// between two accumulations into the same registers, >= 18 wait states before a VALU read; CDNA3 ISA 4.5 / 7.x hazard tables).
__device__ __forceinline__ void mfma16_bagpr_first(f32x4& acc, const u32x4& a, const u32x4& b) {      // acc = a x b (C = 0)
#if FT_OPFMT == 1
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
#else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
#endif
}
__device__ __forceinline__ void mfma16_bagpr(f32x4& acc, const u32x4& a, const u32x4& b) {
#if FT_OPFMT == 1
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
#else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
#endif
}

// ... the same behind two wait states: for an A operand that VALU instructions (the DPP row shifts of member<>) have just written -- the
// compiler pads a builtin MFMA itself (s_nop 0 / 1 in its output), an asm statement gets no such care
__device__ __forceinline__ void mfma16_bagpr_nop(f32x4& acc, const u32x4& a, const u32x4& b) {
#if FT_OPFMT == 1
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
#endif
}


}  // namespace
