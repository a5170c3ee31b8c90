// bf16-image GEMM for the large projections of the training step (FT_BF16 mode, batch == 1, caller-provided workspace).
//
// The fp32-staging kernel of gemm.hip rounds its operands to bf16 on the way into LDS, so every k-step moves 4-byte
// operands through L2 -> VGPR -> cvt -> ds_write: 125 B/clk/CU at the full MFMA rate, twice what the L2 delivers, and the
// transposed operands of the weight gradients pay a register transpose on top.  Here the rounding happens ONCE per
// operand in a streaming pre-pass that also puts the reduction dimension innermost and zero-pads to whole tiles:
//     A(m,k) -> Aimg [ceil128(M)][ceil64(K)] bf16,   B(k,n) -> Bimg [ceil128(N)][ceil64(K)] bf16
// (three source layouts: k contiguous, row contiguous = tiled transpose through LDS, generic strides), and the GEMM
// proper is an "NT" kernel with no bounds checks in its main loop:
//   * 128x128 tile, 32-wide k stages (64 optional), 4 waves (2x2, 64x64 per wave = 4x4 MFMA 16x16x32 tiles per stage),
//   * operands go global -> LDS by `global_load_lds_dwordx4` (no VGPR staging, no ds_write pass), two LDS stages so the
//     DMA of step t+1 is in flight under the MFMAs of step t, one barrier per step, 4 workgroups per CU (32 KiB LDS each),
//   * the LDS image is a sequence of 1 KiB [16 rows][32 k] sub-tiles = exactly one wave-wide DMA each; the DMA writes
//     lane-linear, so the bank swizzle is applied to the SOURCE address: LDS slot(row, kpart) = row*4 + (kpart ^ ((row>>2)&2)),
//     which makes the four 16-lane service groups of ds_read_b128 ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte slots.
// Same epilogue contract as ft_gemm (alpha, beta, bias, activation, optional atomic split-K).
// Rounding is identical to the staging kernel (RNE to bf16, fp32 accumulate); only the summation order differs.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TB = 128;          // tile rows / cols
constexpr int KS = 64;           // k elements per step

inline size_t up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------------------------
// operand images
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    return o;
}

// src(r,k) = src[r*sr + k]   (k contiguous)
__global__ __launch_bounds__(256) void img_rows_k(const float* __restrict__ src, long sr, int R, int K,
                                                  unsigned short* __restrict__ dst, int Rp, int Kp, int vec) {
    const int kq = Kp >> 3;
    const size_t total = (size_t)Rp * kq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / kq), k = (int)(i % kq) * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r < R && k < K) {
            const float* p = src + (size_t)r * sr + k;
            if (vec && k + 7 < K) {
                const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k + e < K) v[e] = p[e];
            }
        }
        *reinterpret_cast<uint4*>(dst + (size_t)r * Kp + k) = pack8(v);
    }
}

// src(r,k) = src[k*sk + r]   (row dim contiguous): 64 x 64 tiled transpose through LDS.  grid (Rp/64, Kp/64)
__global__ __launch_bounds__(256) void img_tr_k(const float* __restrict__ src, long sk, int R, int K,
                                                unsigned short* __restrict__ dst, int Kp, int vec) {
    __shared__ float t[64][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int rq = (tid & 15) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = (tid >> 4) + 16 * j, k = k0 + kk, r = r0 + rq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && r < R) {
            const float* p = src + (size_t)k * sk + r;
            if (vec && r + 3 < R) v = *reinterpret_cast<const float4*>(p);
            else {
                v.x = p[0];
                if (r + 1 < R) v.y = p[1];
                if (r + 2 < R) v.z = p[2];
                if (r + 3 < R) v.w = p[3];
            }
        }
        t[kk][rq] = v.x; t[kk][rq + 1] = v.y; t[kk][rq + 2] = v.z; t[kk][rq + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tid + 256 * j, row = c >> 3, kc = (c & 7) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t[kc + e][row];
        *reinterpret_cast<uint4*>(dst + (size_t)(r0 + row) * Kp + k0 + kc) = pack8(v);
    }
}

// generic strides
__global__ __launch_bounds__(256) void img_generic_k(const float* __restrict__ src, long sr, long sk, int R, int K,
                                                     unsigned short* __restrict__ dst, int Rp, int Kp) {
    const int kq = Kp >> 3;
    const size_t total = (size_t)Rp * kq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / kq), k = (int)(i % kq) * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r < R)
#pragma unroll
            for (int e = 0; e < 8; ++e) if (k + e < K) v[e] = src[(size_t)r * sr + (size_t)(k + e) * sk];
        *reinterpret_cast<uint4*>(dst + (size_t)r * Kp + k) = pack8(v);
    }
}

void make_image(const float* src, long sr, long sk, int R, int K, unsigned short* dst, int Rp, int Kp, hipStream_t st) {
    const bool al = reinterpret_cast<uintptr_t>(src) % 16 == 0;
    const size_t chunks = (size_t)Rp * (Kp >> 3);
    const int blocks = (int)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
    if (sk == 1) {
        hipLaunchKernelGGL(img_rows_k, dim3(blocks), dim3(256), 0, st, src, sr, R, K, dst, Rp, Kp, (al && sr % 4 == 0) ? 1 : 0);
    } else if (sr == 1) {
        hipLaunchKernelGGL(img_tr_k, dim3(Rp / 64, Kp / 64), dim3(256), 0, st, src, sk, R, K, dst, Kp, (al && sk % 4 == 0) ? 1 : 0);
    } else {
        hipLaunchKernelGGL(img_generic_k, dim3(blocks), dim3(256), 0, st, src, sr, sk, R, K, dst, Rp, Kp);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------------------------
struct BfP {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, Kp;
    long ldc;
    float alpha, beta;
    int act, gx, gy, splits, ksteps;        // tile grid, split-K factor, k-steps (of the kernel's KSTEP) per split
    int vec_c;                              // C rows are 16-byte aligned: float4 epilogue
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// KSTEP = k elements per LDS stage (32 or 64), two stages; MINB = workgroups per CU the register budget is held to.
template <int KSTEP, int MINB, bool SPLIT>
__global__ __launch_bounds__(256, MINB) void gemm_bf16_nt(BfP p) {
    constexpr int KH = KSTEP / 32;                     // 32-wide k-halves per stage
    constexpr int STG = 2 * 8 * KH * 1024;             // bytes per stage: (A + B) x 8 row groups x KH sub-tiles of 1 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    {   // XCD-aware order (workgroup L runs on XCD L % 8): every XCD gets a contiguous run of tiles, x fastest
        const int total = p.gx * p.gy, q = total >> 3, r = total & 7;
        const int xcd = tile & 7, idx = tile >> 3;
        tile = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    const int m0 = (tile / p.gx) * TB, n0 = (tile % p.gx) * TB;
    const int nk = p.Kp / KSTEP;
    const int t0 = blockIdx.y * p.ksteps;
    const int t1 = (t0 + p.ksteps < nk) ? t0 + p.ksteps : nk;

    // DMA source of this lane: LDS slot `lane` of a [16][32] sub-tile holds (row lane>>2, k-part (lane&3) ^ swz(row))
    const int srow = lane >> 2, skp = (lane & 3) ^ ((lane >> 4) & 2);
    const unsigned short* ga = p.A + (size_t)(m0 + wave * 32 + srow) * p.Kp + skp * 8;
    const unsigned short* gb = p.B + (size_t)(n0 + wave * 32 + srow) * p.Kp + skp * 8;
    const size_t rstep = (size_t)16 * p.Kp;
    auto issue = [&](int stage, int t) {
        unsigned char* sa = smem + stage * STG;
        unsigned char* sb = sa + STG / 2;
        const size_t ko = (size_t)t * KSTEP;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int sub = (kh * 8 + wave * 2 + g) << 10;
                __builtin_amdgcn_global_load_lds((glb_void*)(ga + g * rstep + ko + kh * 32), (lds_void*)(sa + sub), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void*)(gb + g * rstep + ko + kh * 32), (lds_void*)(sb + sub), 16, 0, 0);
            }
    };

    // !SPLIT: acc[i][j] holds C^T: lane (li, kg), register r  <->  C[m = i*16 + li][n = j*16 + kg*4 + r]  (operands swapped
    //         in the MFMA so that a lane owns four CONSECUTIVE output columns: float4 stores / loads in the epilogue)
    //  SPLIT: natural order, register r <-> C[m = i*16 + kg*4 + r][n = j*16 + li]: one atomic instruction then covers 16
    //         consecutive columns of 4 rows (4 cache lines) instead of 4 columns of 16 rows (measured 20-60 % faster)
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int rslot = (li * 4 + (kg ^ ((li >> 2) & 2))) * 16;      // this lane's fragment slot inside a sub-tile
    if (t0 < t1) issue(0, t0);
    for (int t = t0; t < t1; ++t) {
        const int stage = (t - t0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my DMAs of step t have landed
        __syncthreads();                                            // ... everyone's have; stage^1 is no longer being read
        if (t + 1 < t1) issue(stage ^ 1, t + 1);
        const unsigned char* sa = smem + stage * STG + rslot;
        const unsigned char* sb = sa + STG / 2;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + ((kh * 8 + wm * 4 + i) << 10));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + ((kh * 8 + wn * 4 + j) << 10));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                }
        }
    }

    if constexpr (SPLIT) {          // C was zeroed (beta == 0) or holds the addend (beta == 1)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + kg * 4 + r;
                if (row >= p.M) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wn * 64 + j * 16 + li;
                    if (col >= p.N) continue;
                    float v = p.alpha * acc[i][j][r];
                    if (p.bias && blockIdx.y == 0) v += p.bias[col];
                    atomicAdd(p.C + (long)row * p.ldc + col, v);
                }
            }
        return;
    }

    const bool vec = p.vec_c != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 64 + i * 16 + li;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + kg * 4;
            if (col >= p.N) continue;
            float* cp = p.C + (long)row * p.ldc + col;
            float v[4] = {p.alpha * acc[i][j][0], p.alpha * acc[i][j][1], p.alpha * acc[i][j][2], p.alpha * acc[i][j][3]};
            const int nv = (p.N - col < 4) ? p.N - col : 4;
            const bool full = vec && nv == 4;
            if (p.beta != 0.f) {
                if (full) { const float4 c = *reinterpret_cast<const float4*>(cp); v[0] += p.beta * c.x; v[1] += p.beta * c.y; v[2] += p.beta * c.z; v[3] += p.beta * c.w; }
                else for (int r = 0; r < nv; ++r) v[r] += p.beta * cp[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < nv && p.bias) v[r] += p.bias[col + r];
                if (p.act == FT_ACT_TANH) v[r] = tanhf_(v[r]);
                else if (p.act == FT_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
                else if (p.act == FT_ACT_SIGMOID) v[r] = sigmoidf_(v[r]);
            }
            if (full) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int r = 0; r < nv; ++r) cp[r] = v[r];
        }
    }
}

bool qualifies(const ft_gemm_args* a) {
    return a->mode == FT_BF16 && a->batch == 1 && a->M >= 32 && a->N >= 32 && a->K >= 16 &&
           (double)a->M * a->N * a->K >= (double)(1 << 20);
}

}  // namespace

extern "C" size_t ft_gemm_workspace_bytes(const ft_gemm_args* a) {
    if (!a || !qualifies(a)) return 0;
    const size_t Mp = up(a->M, TB), Np = up(a->N, TB), Kp = up(a->K, KS);
    return up(Mp * Kp * 2, 256) + up(Np * Kp * 2, 256);
}

// returns 1 when the call was taken by this path, 0 when the caller should use the fp32-staging kernel, < 0 on error
int ftint_gemm_bf16(const ft_gemm_args* a, hipStream_t st) {
    if (!qualifies(a) || !a->work) return 0;
    const size_t need = ft_gemm_workspace_bytes(a);
    if (a->work_bytes < need) return 0;
    if (reinterpret_cast<uintptr_t>(a->work) % 256 != 0) return ft_fail(FT_EINVAL, "ft_gemm: work must be 256-byte aligned");
    const int Mp = (int)up(a->M, TB), Np = (int)up(a->N, TB), Kp = (int)up(a->K, KS);
    unsigned short* Aimg = reinterpret_cast<unsigned short*>(a->work);
    unsigned short* Bimg = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(a->work) + up((size_t)Mp * Kp * 2, 256));
    make_image(a->A, a->sAm, a->sAk, a->M, a->K, Aimg, Mp, Kp, st);
    make_image(a->B, a->sBn, a->sBk, a->N, a->K, Bimg, Np, Kp, st);

    BfP p;
    p.A = Aimg; p.B = Bimg; p.C = a->C; p.bias = a->bias;
    p.M = a->M; p.N = a->N; p.Kp = Kp; p.ldc = a->ldc;
    p.alpha = a->alpha; p.beta = a->beta; p.act = a->act;
    p.gx = Np / TB; p.gy = Mp / TB;
    static const int variant = [] { const char* e = getenv("FT_GEMM_BF16_VARIANT"); return e ? atoi(e) : 0; }();
    const int kstep = (variant == 0) ? 32 : 64;        // default: 32-wide stages, 4 workgroups/CU (measured 3-15 % faster than 64 / 2)
    const int nk = Kp / kstep;
    p.vec_c = (reinterpret_cast<uintptr_t>(a->C) % 16 == 0 && a->ldc % 4 == 0) ? 1 : 0;
    const bool can_split = (a->flags & FT_GEMM_SPLITK) && a->act == FT_ACT_NONE && (a->beta == 0.f || a->beta == 1.f) && a->K >= 2048;
    const long tiles = (long)p.gx * p.gy;
    long s = 1;
    if (can_split && tiles < 512) {
        s = 768 / tiles;
        const long smax = a->K / 512;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    p.ksteps = cdiv(nk, s);
    p.splits = cdiv(nk, p.ksteps);
    if (p.splits > 1 && a->beta == 0.f)
        FT_CHECK_HIP(hipMemset2DAsync(a->C, sizeof(float) * a->ldc, 0, sizeof(float) * a->N, a->M, st));
    const dim3 grid(p.gx * p.gy, p.splits);
    if (variant == 0) {
        if (p.splits > 1) hipLaunchKernelGGL((gemm_bf16_nt<32, 4, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_nt<32, 4, false>), grid, dim3(256), 0, st, p);
    } else {
        if (p.splits > 1) hipLaunchKernelGGL((gemm_bf16_nt<64, 2, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_nt<64, 2, false>), grid, dim3(256), 0, st, p);
    }
    FT_CHECK_LAUNCH();
    return 1;
}
