// Shared device/host helpers for libflowtron_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/flowtron_hip.h"

extern thread_local char g_ft_err[512];

static inline int ft_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ft_err, sizeof(g_ft_err), fmt, ap);
    va_end(ap);
    return code;
}

#define FT_CHECK_ARG(cond)                                                            \
    do {                                                                              \
        if (!(cond)) return ft_fail(FT_EINVAL, "%s: invalid argument: %s", __func__, #cond); \
    } while (0)

#define FT_CHECK_LAUNCH()                                                             \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess)                                                         \
            return ft_fail(FT_EHIP, "%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
    } while (0)

#define FT_CHECK_HIP(expr)                                                            \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return ft_fail(FT_EHIP, "%s: %s: %s", __func__, #expr, hipGetErrorString(e_)); \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

// 16-bit MFMA operand format of this translation unit.  The operand-typed files (gemm, gemm_bf16, lstm, lstm2, lstm_persist)
// are compiled TWICE: FT_OPFMT 0 = bf16 (FT_BF16), 1 = fp16 (FT_F16: v_mfma_f32_16x16x32_f16, the reference's fp16 AMP
// configuration).  In the fp16 objects every format-dependent extern "C" entry carries the suffix _f16 (FT_OPNAME); entries
// with a `mode` argument dispatch to their twin themselves.  Everything else -- images, fragments, hand-off granules -- only
// moves opaque 16-bit payloads, so the two builds share every line of code but the two functions below.
#ifndef FT_OPFMT
#define FT_OPFMT 0
#endif
#if FT_OPFMT == 1
#define FT_OPNAME(x) x##_f16
#define FT_OP16 FT_F16
#else
#define FT_OPNAME(x) x
#define FT_OP16 FT_BF16
#endif

// round-to-nearest-even fp32 -> 16-bit operand through the hardware converters (v_cvt_pk_bf16_f32 / v_cvt_f16_f32).
// fp16 saturates to +-inf beyond 65504: a scaled gradient that overflows is what GradScaler's inf check is for.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_hw;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_hw;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu); }
#if FT_OPFMT == 1
// one instruction for every fp32 -> fp16 rounding in the library: hipcc lowers a vector convert to v_cvt_pk_f16_f32 and a scalar
// one to v_cvt_f16_f32, and on gfx950 the two do NOT round ties alike (measured: the persistent and the launch-per-step
// recurrences, which differ only in which of the two the compiler picked, diverged on ~1 value in 8192) -- v_cvt_f16_f32 is
// the RNE one (tests/test_gpu_ops.py compares the images with torch's cast bit for bit)
__device__ __forceinline__ unsigned int cvt_f16_bits(float f) {
    unsigned int r;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(f));
    return r & 0xffffu;
}
#endif
__device__ __forceinline__ unsigned int pack_op16x2(float lo, float hi) {
#if FT_OPFMT == 1
    return cvt_f16_bits(lo) | (cvt_f16_bits(hi) << 16);
#else
    return pack_bf16x2(lo, hi);
#endif
}
__device__ __forceinline__ unsigned short f2op16(float f) {
#if FT_OPFMT == 1
    return (unsigned short)cvt_f16_bits(f);
#else
    return f2bf(f);
#endif
}
__device__ __forceinline__ float op16_to_f(unsigned int bits16) {                 // exact widening of one 16-bit operand
#if FT_OPFMT == 1
    return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16);
#else
    return __builtin_bit_cast(float, bits16 << 16);
#endif
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {          // 16x16x32, fp32 accumulate
#if FT_OPFMT == 1
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// accurate-enough fp32 tanh: 1 - 2/(exp(2x)+1); exact limits at +-inf, |err| ~ 1e-7
__device__ __forceinline__ float tanhf_(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

// LSTM cell update shared by every recurrence kernel (lstm.hip, lstm2.hip, lstm_persist.hip), with the product / FMA order
// pinned (the default -ffp-contract=fast may fuse `f*c + i*g` either way in different kernels, and a 1-ulp difference is
// amplified by the recurrence): kernels that feed it the same pre-activations produce bit-identical states.
//   FAST = false (fp32 parity mode): libm expf / tanhf.
//   FAST = true  (bf16 / fp16 operand modes): v_exp_f32 + v_rcp_f32 forms, |abs err| ~ 1e-7 -- two orders below the operand
//   rounding -- which takes ~100 instructions off the per-step critical path.
template <bool FAST>
__device__ __forceinline__ void lstm_cell(const float (&pre)[4], float c_old, float& ig, float& fg, float& gg, float& og,
                                          float& c_new, float& h_new) {
    if constexpr (FAST) {
        constexpr float L2E = 1.4426950408889634f;
        ig = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * pre[0]));
        fg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * pre[1]));
        gg = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * L2E * pre[2]) + 1.f);
        og = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-L2E * pre[3]));
        c_new = __fmaf_rn(fg, c_old, __fmul_rn(ig, gg));
        h_new = __fmul_rn(og, 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * L2E * c_new) + 1.f));
    } else {
        ig = 1.f / (1.f + expf(-pre[0]));
        fg = 1.f / (1.f + expf(-pre[1]));
        gg = tanhf(pre[2]);
        og = 1.f / (1.f + expf(-pre[3]));
        c_new = __fmaf_rn(fg, c_old, __fmul_rn(ig, gg));
        h_new = __fmul_rn(og, tanhf(c_new));
    }
}

// LSTM cell backward, product order pinned like lstm_cell:
//   dc = dh o (1 - tanh^2 c_t) + dc_carry ;  carry' = dc f ;  da_i = dc g i(1-i) ; da_f = dc c_prev f(1-f) ;
//   da_g = dc i (1-g^2) ; da_o = dh tanh(c_t) o(1-o)
template <bool FAST>
__device__ __forceinline__ void lstm_cell_bwd(float dh, float dc_carry, float ig, float fg, float gg, float og, float c_t,
                                              float c_prev, float (&da)[4], float& carry_out) {
    float tc;
    if constexpr (FAST) tc = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * 1.4426950408889634f * c_t) + 1.f);
    else tc = tanhf(c_t);
    const float dc = __fmaf_rn(__fmul_rn(dh, og), __fmaf_rn(-tc, tc, 1.f), dc_carry);
    carry_out = __fmul_rn(dc, fg);
    da[0] = __fmul_rn(__fmul_rn(__fmul_rn(dc, gg), ig), 1.f - ig);
    da[1] = __fmul_rn(__fmul_rn(__fmul_rn(dc, c_prev), fg), 1.f - fg);
    da[2] = __fmul_rn(__fmul_rn(dc, ig), __fmaf_rn(-gg, gg, 1.f));
    da[3] = __fmul_rn(__fmul_rn(__fmul_rn(dh, tc), og), 1.f - og);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// gemm_bf16.hip: 1 = handled, 0 = not applicable (use the staging kernel), < 0 = error
int FT_OPNAME(ftint_gemm_bf16)(const ft_gemm_args* a, hipStream_t st);

// cumm_fused.hip (one object per operand format): the fused cumulative-attention frames behind ft_cumm_attn_fwd / _bwd
#define FT_CUMMF_DECL(sfx)                                                                                                              \
    int ftint_cummf_supported##sfx(const ft_cumm_attn_args* a);                                                                        \
    void ftint_cummf_debug_prof##sfx(void* dev_buf);                                                                                   \
    size_t ftint_cummf_workspace_bytes##sfx(int T, int L, int B, int E, int A, int backward);                                          \
    int ftint_cummf_fwd##sfx(const ft_cumm_attn_args* a, hipStream_t st);                                                              \
    int ftint_cummf_bwd##sfx(const ft_cumm_attn_args* a, const float* dctx, const float* dattn, const float* dlogprob, float* dQ,      \
                             float* dV, float* dtext, float* dw_key, float* dv, float* dw1, float* db1, float* dw2, float* db2, hipStream_t st);
FT_CUMMF_DECL()
FT_CUMMF_DECL(_f16)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
