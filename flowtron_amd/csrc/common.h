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

// round-to-nearest-even fp32 -> bf16 through the gfx950 hardware converter (v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// accurate-enough fp32 tanh: 1 - 2/(exp(2x)+1); exact limits at +-inf, |err| ~ 1e-7
__device__ __forceinline__ float tanhf_(float x) {
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
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
int ftint_gemm_bf16(const ft_gemm_args* a, hipStream_t st);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
