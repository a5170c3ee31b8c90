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

// ---- the same two passes with 16 row lanes x 16 channel quads per workgroup (C % 4 == 0, 16-byte aligned rows; round 6): a thread
// walks len / 16 rows with float4 loads instead of len / 4 rows with scalar ones -- the loops are chains of dependent L2 round trips
// (one workgroup per (sample, 64 channels): 256 workgroups), so a quarter of the trips is most of the time (48 -> us forward).
__device__ __forceinline__ float4 red16(float4 v, float4 (*red)[16], int rl, int cq) {
    __syncthreads();
    red[rl][cq] = v;
    __syncthreads();
    float4 s = red[0][cq];
#pragma unroll
    for (int r = 1; r < 16; ++r) { const float4 t = red[r][cq]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    return s;
}
#define F4(p) (*reinterpret_cast<const float4*>(p))

__global__ __launch_bounds__(256) void instnorm_relu_fwd4_k(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ keep,
                                                            const int* __restrict__ lens, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int L, int B, int C, float eps) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cq * 4;
    const bool cv = c < C;
    const int len = min(lens[b], L);
    const long rs = (long)B * C;
    const float* xp = x + (long)b * C + c;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (cv) for (int l = rl; l < len; l += 16) { const float4 v = F4(xp + l * rs); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    float4 mean = red16(s, red, rl, cq);
    const float il = 1.f / (float)len;
    mean.x *= il; mean.y *= il; mean.z *= il; mean.w *= il;
    float4 q = {0.f, 0.f, 0.f, 0.f};
    if (cv) for (int l = rl; l < len; l += 16) {
        const float4 v = F4(xp + l * rs);
        const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
    const float4 var = red16(q, red, rl, cq);
    const float4 rstd = {1.0f / sqrtf(var.x * il + eps), 1.0f / sqrtf(var.y * il + eps), 1.0f / sqrtf(var.z * il + eps), 1.0f / sqrtf(var.w * il + eps)};
    if (!cv) return;
    if (rl == 0) { *reinterpret_cast<float4*>(mean_out + (long)b * C + c) = mean; *reinterpret_cast<float4*>(rstd_out + (long)b * C + c) = rstd; }
    const float4 g = F4(gamma + c), be = F4(beta + c);
    float* yp = y + (long)b * C + c;
    const float* kp = keep ? keep + (long)b * C + c : nullptr;
    for (int l = rl; l < L; l += 16) {
        float4 o = {0.f, 0.f, 0.f, 0.f};
        if (l < len) {
            const float4 v = F4(xp + l * rs);
            o.x = fmaxf((v.x - mean.x) * rstd.x * g.x + be.x, 0.f); o.y = fmaxf((v.y - mean.y) * rstd.y * g.y + be.y, 0.f);
            o.z = fmaxf((v.z - mean.z) * rstd.z * g.z + be.z, 0.f); o.w = fmaxf((v.w - mean.w) * rstd.w * g.w + be.w, 0.f);
            if (kp) { const float4 k = F4(kp + l * rs); o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w; }
        }
        *reinterpret_cast<float4*>(yp + l * rs) = o;
    }
}

__global__ __launch_bounds__(256) void instnorm_relu_bwd4_k(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ keep, const int* __restrict__ lens,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int L, int B, int C) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cq * 4;
    const bool cv = c < C;
    const int len = min(lens[b], L);
    const long rs = (long)B * C;
    const long off = (long)b * C + c;
    const float4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float4 mean = cv ? F4(mean_in + off) : z4, rstd = cv ? F4(rstd_in + off) : z4, gm = cv ? F4(gamma + c) : z4;
    auto gof = [&](int l) {                                        // upstream gradient through the ReLU and the dropout mask
        const float4 yv = F4(y + off + l * rs), d = F4(dy + off + l * rs);
        float4 g = {yv.x > 0.f ? d.x : 0.f, yv.y > 0.f ? d.y : 0.f, yv.z > 0.f ? d.z : 0.f, yv.w > 0.f ? d.w : 0.f};
        if (keep) { const float4 k = F4(keep + off + l * rs); g.x *= k.x; g.y *= k.y; g.z *= k.z; g.w *= k.w; }
        return g;
    };
    auto xhat = [&](int l) {
        const float4 v = F4(x + off + l * rs);
        return float4{(v.x - mean.x) * rstd.x, (v.y - mean.y) * rstd.y, (v.z - mean.z) * rstd.z, (v.w - mean.w) * rstd.w};
    };
    float4 sg = z4, sgx = z4;
    if (cv)
        for (int l = rl; l < len; l += 16) {
            const float4 g = gof(l), xh = xhat(l);
            sg.x += g.x; sg.y += g.y; sg.z += g.z; sg.w += g.w;
            sgx.x += g.x * xh.x; sgx.y += g.y * xh.y; sgx.z += g.z * xh.z; sgx.w += g.w * xh.w;
        }
    const float4 Sg = red16(sg, red, rl, cq);
    const float4 Sgx = red16(sgx, red, rl, cq);
    if (!cv) return;
    if (rl == 0) {
        atomicAdd(dgamma + c, Sgx.x); atomicAdd(dgamma + c + 1, Sgx.y); atomicAdd(dgamma + c + 2, Sgx.z); atomicAdd(dgamma + c + 3, Sgx.w);
        atomicAdd(dbeta + c, Sg.x); atomicAdd(dbeta + c + 1, Sg.y); atomicAdd(dbeta + c + 2, Sg.z); atomicAdd(dbeta + c + 3, Sg.w);
    }
    const float il = 1.f / (float)len;
    for (int l = rl; l < L; l += 16) {
        float4 o = z4;
        if (l < len) {
            const float4 g = gof(l), xh = xhat(l);
            o.x = gm.x * rstd.x * (g.x - Sg.x * il - xh.x * Sgx.x * il); o.y = gm.y * rstd.y * (g.y - Sg.y * il - xh.y * Sgx.y * il);
            o.z = gm.z * rstd.z * (g.z - Sg.z * il - xh.z * Sgx.z * il); o.w = gm.w * rstd.w * (g.w - Sg.w * il - xh.w * Sgx.w * il);
        }
        *reinterpret_cast<float4*>(dx + off + l * rs) = o;
    }
}
#undef F4

}  // namespace

extern "C" int ft_instnorm_relu_fwd(const float* x, const float* gamma, const float* beta, const float* keep,
                                    const int32_t* lens, float* y, float* mean, float* rstd,
                                    int L, int B, int C, float eps, void* stream) {
    FT_CHECK_ARG(x && gamma && beta && lens && y && mean && rstd && L >= 1 && B >= 1 && C >= 1);
    FT_CHECK_ARG(B <= 65535);
    const bool q4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(keep) |
                                    reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(mean) |
                                    reinterpret_cast<uintptr_t>(rstd)) % 16 == 0);
    if (q4) hipLaunchKernelGGL(instnorm_relu_fwd4_k, dim3(cdiv(C, 64), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                               x, gamma, beta, keep, lens, y, mean, rstd, L, B, C, eps);
    else hipLaunchKernelGGL(instnorm_relu_fwd_k, dim3(cdiv(C, 64), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
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
    const bool q4 = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
                                    reinterpret_cast<uintptr_t>(keep) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(mean) |
                                    reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(dx)) % 16 == 0);
    if (q4) hipLaunchKernelGGL(instnorm_relu_bwd4_k, dim3(cdiv(C, 64), B), dim3(256), 0, st,
                               x, y, dy, gamma, keep, lens, mean, rstd, dx, dgamma, dbeta, L, B, C);
    else hipLaunchKernelGGL(instnorm_relu_bwd_k, dim3(cdiv(C, 64), B), dim3(256), 0, st,
                            x, y, dy, gamma, keep, lens, mean, rstd, dx, dgamma, dbeta, L, B, C);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
