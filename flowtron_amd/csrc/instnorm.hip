// Length-masked instance norm + ReLU (+ dropout keep-mask) over time-major [L,B,C]
// activations (encoder conv stack, reference flowtron.py:53-92 + :502).
// One workgroup per (sample b, 64-channel slab): 4 row-lanes x 64 channels, so every
// global access is a 256 B coalesced segment along C; the whole [len_b, 64] slab is
// re-read from L2 for the three passes (mean, biased variance, normalise).
#include "common.h"

namespace {

__device__ __forceinline__ float red4(float v, float (*red)[64], int rl, int cl) {
    __syncthreads();
    red[rl][cl] = v;
    __syncthreads();
    return red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

__global__ __launch_bounds__(256) void instnorm_relu_fwd_k(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ keep,
                                                           const int* __restrict__ lens, float* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int L, int B, int C, float eps) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cl;
    const bool cv = c < C;
    const int len = min(lens[b], L);
    const long rs = (long)B * C;
    const float* xp = x + (long)b * C + c;
    float s = 0.f;
    if (cv) for (int l = rl; l < len; l += 4) s += xp[l * rs];
    const float mean = red4(s, red, rl, cl) / (float)len;
    float q = 0.f;
    if (cv) for (int l = rl; l < len; l += 4) { const float d = xp[l * rs] - mean; q += d * d; }
    const float var = red4(q, red, rl, cl) / (float)len;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!cv) return;
    if (rl == 0) { mean_out[(long)b * C + c] = mean; rstd_out[(long)b * C + c] = rstd; }
    const float g = gamma[c], be = beta[c];
    float* yp = y + (long)b * C + c;
    const float* kp = keep ? keep + (long)b * C + c : nullptr;
    for (int l = rl; l < L; l += 4) {
        float v = 0.f;
        if (l < len) {
            v = fmaxf((xp[l * rs] - mean) * rstd * g + be, 0.f);
            if (kp) v *= kp[l * rs];
        }
        yp[l * rs] = v;
    }
}

__global__ __launch_bounds__(256) void instnorm_relu_bwd_k(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ gamma,
                                                           const float* __restrict__ keep, const int* __restrict__ lens,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int L, int B, int C) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cl;
    const bool cv = c < C;
    const int len = min(lens[b], L);
    const long rs = (long)B * C;
    const long off = (long)b * C + c;
    const float mean = cv ? mean_in[off] : 0.f, rstd = cv ? rstd_in[off] : 0.f;
    const float gm = cv ? gamma[c] : 0.f;
    float sg = 0.f, sgx = 0.f;
    if (cv)
        for (int l = rl; l < len; l += 4) {
            float g = (y[off + l * rs] > 0.f) ? dy[off + l * rs] : 0.f;
            if (keep) g *= keep[off + l * rs];
            const float xh = (x[off + l * rs] - mean) * rstd;
            sg += g; sgx += g * xh;
        }
    const float Sg = red4(sg, red, rl, cl);
    const float Sgx = red4(sgx, red, rl, cl);
    if (!cv) return;
    if (rl == 0) { atomicAdd(dgamma + c, Sgx); atomicAdd(dbeta + c, Sg); }
    const float mg = Sg / (float)len, mgx = Sgx / (float)len;
    for (int l = rl; l < L; l += 4) {
        float v = 0.f;
        if (l < len) {
            float g = (y[off + l * rs] > 0.f) ? dy[off + l * rs] : 0.f;
            if (keep) g *= keep[off + l * rs];
            const float xh = (x[off + l * rs] - mean) * rstd;
            v = gm * rstd * (g - mg - xh * mgx);
        }
        dx[off + l * rs] = v;
    }
}

}  // namespace

extern "C" int ft_instnorm_relu_fwd(const float* x, const float* gamma, const float* beta, const float* keep,
                                    const int32_t* lens, float* y, float* mean, float* rstd,
                                    int L, int B, int C, float eps, void* stream) {
    FT_CHECK_ARG(x && gamma && beta && lens && y && mean && rstd && L >= 1 && B >= 1 && C >= 1);
    FT_CHECK_ARG(B <= 65535);
    hipLaunchKernelGGL(instnorm_relu_fwd_k, dim3(cdiv(C, 64), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, gamma, beta, keep, lens, y, mean, rstd, L, B, C, eps);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int ft_instnorm_relu_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* keep,
                                    const int32_t* lens, const float* mean, const float* rstd,
                                    float* dx, float* dgamma, float* dbeta, int L, int B, int C, void* stream) {
    FT_CHECK_ARG(x && y && dy && gamma && lens && mean && rstd && dx && dgamma && dbeta && L >= 1 && B >= 1 && C >= 1);
    FT_CHECK_ARG(B <= 65535);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FT_CHECK_HIP(hipMemsetAsync(dgamma, 0, sizeof(float) * C, st));
    FT_CHECK_HIP(hipMemsetAsync(dbeta, 0, sizeof(float) * C, st));
    hipLaunchKernelGGL(instnorm_relu_bwd_k, dim3(cdiv(C, 64), B), dim3(256), 0, st,
                       x, y, dy, gamma, keep, lens, mean, rstd, dx, dgamma, dbeta, L, B, C);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
