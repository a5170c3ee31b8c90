// Strided batched GEMM with fused epilogue on the gfx950 matrix cores.
//   C = act(alpha * A.B + beta * C + bias)
// Tile 128x128x32 with 4 waves (2x2, each wave 64x64 = 4x4 MFMA tiles of 16x16).  (A 256x256x32 / 512-thread tile was
// built and measured 4 ms/step slower on the training step -- DESIGN.md "tried and measured NOT to help" -- and removed.)
// MODE 0: fp32 operands, v_mfma_f32_16x16x4_f32 (exact fp32).
// MODE 1: operands rounded to bf16 while being staged into LDS,
// v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Global operands are always fp32.
//
// Operand staging handles three global layouts per operand (chosen on the host):
//   1 = reduction dim contiguous (float4 along k),
//   2 = row dim contiguous (4x4 register transpose, float4 along the row dim),
//   0 = generic strides (scalar loads).
// LDS image is always [row][k] with k innermost so the MFMA read side is uniform.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDF = 36;   // fp32 LDS row stride in floats  (144 B, 16 B aligned)
constexpr int LDH = 40;   // bf16 LDS row stride in halves  (80 B, 16 B aligned)

struct GemmP {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long sAm, sAk, sBk, sBn, ldc, bsA, bsB, bsC;
    float alpha, beta;
    int act, amode, bmode;
    int gx, gy, splits, kchunk;   // tile grid, split-K factor and K elements per split (multiple of BK)
};

// Load one (NT/2) x 32 operand tile (rows r0.., reduction k0..) into 16 registers/thread (NT threads).
template <int LMODE, int NT>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, long sr, long sk, int R, int K,
                                          int r0, int k0, int tid, float (&reg)[16]) {
    if constexpr (LMODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int idx = tid + NT * j;
            int r = r0 + (idx >> 3), k = k0 + (idx & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const float* p = X + (long)r * sr + k;
                if (k + 3 < K) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (k < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                }
            }
            reg[j * 4 + 0] = v.x; reg[j * 4 + 1] = v.y; reg[j * 4 + 2] = v.z; reg[j * 4 + 3] = v.w;
        }
    } else if constexpr (LMODE == 2) {
        constexpr int RQ = NT / 8;                      // row quads per tile: 32 (128 rows) or 64 (256 rows)
        int rq = tid % RQ, kq = tid / RQ;
        int r = r0 + rq * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            int k = k0 + kq * 4 + kk;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K) {
                const float* p = X + (long)k * sk + r;
                if (r + 3 < R) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (r < R) v.x = p[0];
                    if (r + 1 < R) v.y = p[1];
                    if (r + 2 < R) v.z = p[2];
                }
            }
            // reg[j*4+kk] = element (row rq*4+j, k kq*4+kk)
            reg[0 * 4 + kk] = v.x; reg[1 * 4 + kk] = v.y; reg[2 * 4 + kk] = v.z; reg[3 * 4 + kk] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int idx = tid + NT * j;
            int r = r0 + (idx >> 5), k = k0 + (idx & 31);
            reg[j] = (r < R && k < K) ? X[(long)r * sr + (long)k * sk] : 0.f;
        }
    }
}

template <int MODE, int LMODE, int NT>
__device__ __forceinline__ void store_tile(void* lds, int tid, const float (&reg)[16]) {
    if constexpr (LMODE == 1 || LMODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row, kc;
            if constexpr (LMODE == 1) { int idx = tid + NT * j; row = idx >> 3; kc = (idx & 7) * 4; }
            else { row = (tid % (NT / 8)) * 4 + j; kc = (tid / (NT / 8)) * 4; }
            if constexpr (MODE == 0) {
                float* p = reinterpret_cast<float*>(lds) + row * LDF + kc;
                *reinterpret_cast<float4*>(p) = make_float4(reg[j * 4], reg[j * 4 + 1], reg[j * 4 + 2], reg[j * 4 + 3]);
            } else {
                unsigned short* p = reinterpret_cast<unsigned short*>(lds) + row * LDH + kc;
                uint2 w;
                w.x = pack_op16x2(reg[j * 4], reg[j * 4 + 1]);
                w.y = pack_op16x2(reg[j * 4 + 2], reg[j * 4 + 3]);
                *reinterpret_cast<uint2*>(p) = w;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int idx = tid + NT * j;
            int row = idx >> 5, kc = idx & 31;
            if constexpr (MODE == 0) reinterpret_cast<float*>(lds)[row * LDF + kc] = reg[j];
            else reinterpret_cast<unsigned short*>(lds)[row * LDH + kc] = f2op16(reg[j]);
        }
    }
}

// NT = 256: 128x128 tile, waves 2(M) x 2(N), wave tile 64x64 (TI = 4 x TJ = 4 MFMA tiles)
// (NT stays a parameter of the staging helpers; only NT = 256 is instantiated)
template <int MODE, int AMODE, int BMODE, int NT>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmP p) {
    constexpr int BMT = NT / 2, BNT = NT / 2;               // tile rows / cols
    constexpr int TI = (NT == 256) ? 4 : 8, TJ = 4;
    constexpr int WN = (NT == 256) ? 2 : 4;                  // waves along N
    constexpr int TILE_BYTES = (MODE == 0) ? BMT * LDF * 4 : BMT * LDH * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    void* As = smem;
    void* Bs = smem + TILE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kg = lane >> 4;
    // XCD-aware tile order: workgroup L runs on XCD L % 8 (observed round-robin dispatch); give every XCD a
    // contiguous run of tiles (x fastest) so the tiles that share an A row-panel hit the same 4 MiB L2.
    int tile = blockIdx.x;
    {
        const int total = p.gx * p.gy, q = total >> 3, r = total & 7;
        const int xcd = tile & 7, idx = tile >> 3;
        tile = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    const int m0 = (tile / p.gx) * BMT, n0 = (tile % p.gx) * BNT;
    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
    const long bz = blockIdx.z;
    const float* A = p.A + bz * p.bsA;
    const float* B = p.B + bz * p.bsB;
    float* C = p.C + bz * p.bsC;

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float ra[16], rb[16];
    load_tile<AMODE, NT>(A, p.sAm, p.sAk, p.M, kend, m0, kbeg, tid, ra);
    load_tile<BMODE, NT>(B, p.sBn, p.sBk, p.N, kend, n0, kbeg, tid, rb);

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                         // previous tile fully consumed
        store_tile<MODE, AMODE, NT>(As, tid, ra);
        store_tile<MODE, BMODE, NT>(Bs, tid, rb);
        __syncthreads();
        if (k0 + BK < kend) {                    // prefetch next tile while computing
            load_tile<AMODE, NT>(A, p.sAm, p.sAk, p.M, kend, m0, k0 + BK, tid, ra);
            load_tile<BMODE, NT>(B, p.sBn, p.sBk, p.N, kend, n0, k0 + BK, tid, rb);
        }
        if constexpr (MODE == 0) {
            const float* Af = reinterpret_cast<const float*>(As) + (wm * (TI * 16) + li) * LDF + kg * 4;
            const float* Bf = reinterpret_cast<const float*>(Bs) + (wn * (TJ * 16) + li) * LDF + kg * 4;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 16) {
                float4 a[TI], b[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) a[i] = *reinterpret_cast<const float4*>(Af + i * 16 * LDF + kk);
#pragma unroll
                for (int j = 0; j < TJ; ++j) b[j] = *reinterpret_cast<const float4*>(Bf + j * 16 * LDF + kk);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As) + (wm * (TI * 16) + li) * LDH + kg * 8;
            const unsigned short* Bh = reinterpret_cast<const unsigned short*>(Bs) + (wn * (TJ * 16) + li) * LDH + kg * 8;
            bf16x8 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Ah + i * 16 * LDH);
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bh + j * 16 * LDH);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
    }

    // epilogue: D tile (16x16): col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = m0 + wm * (TI * 16) + i * 16 + kg * 4 + r;
            if (row >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                int col = n0 + wn * (TJ * 16) + j * 16 + li;
                if (col >= p.N) continue;
                float* cp = C + (long)row * p.ldc + col;
                float v = p.alpha * acc[i][j][r];
                if (p.splits > 1) {              // split-K: C was zeroed (beta == 0) or holds the addend (beta == 1)
                    if (p.bias && blockIdx.y == 0) v += p.bias[col];
                    atomicAdd(cp, v);
                    continue;
                }
                if (p.beta != 0.f) v += p.beta * (*cp);
                if (p.bias) v += p.bias[col];
                if (p.act == FT_ACT_TANH) v = tanhf_(v);
                else if (p.act == FT_ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == FT_ACT_SIGMOID) v = sigmoidf_(v);
                *cp = v;
            }
        }
    }
}

template <int MODE, int AMODE>
void launch_b(const GemmP& p, dim3 grid, hipStream_t st) {
    switch (p.bmode) {
        case 1: hipLaunchKernelGGL((gemm_kernel<MODE, AMODE, 1, 256>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<MODE, AMODE, 2, 256>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<MODE, AMODE, 0, 256>), grid, dim3(256), 0, st, p); break;
    }
}
template <int MODE>
void launch_a(const GemmP& p, dim3 grid, hipStream_t st) {
    switch (p.amode) {
        case 1: launch_b<MODE, 1>(p, grid, st); break;
        case 2: launch_b<MODE, 2>(p, grid, st); break;
        default: launch_b<MODE, 0>(p, grid, st); break;
    }
}

// decide the staging path of one operand: rows stride sr, reduction stride sk
int pick_mode(const float* base, long sr, long sk, long bs, int batch) {
    bool al = (reinterpret_cast<uintptr_t>(base) % 16 == 0) && (batch <= 1 || bs % 4 == 0);
    if (sk == 1 && al && sr % 4 == 0) return 1;
    if (sr == 1 && al && sk % 4 == 0) return 2;
    return 0;
}

}  // namespace

#if FT_OPFMT == 0
extern "C" int ft_gemm_f16(const ft_gemm_args* a, void* stream);
#endif
extern "C" int FT_OPNAME(ft_gemm)(const ft_gemm_args* a, void* stream) {
    FT_CHECK_ARG(a != nullptr);
#if FT_OPFMT == 0
    if (a->mode == FT_F16) return ft_gemm_f16(a, stream);        // the fp16-operand build of this file
#endif
    FT_CHECK_ARG(a->A && a->B && a->C);
    FT_CHECK_ARG(a->M >= 0 && a->N >= 0 && a->K >= 0 && a->batch >= 1);
    FT_CHECK_ARG(a->mode == FT_F32 || a->mode == FT_OP16);
    if (a->M == 0 || a->N == 0) return FT_OK;
    FT_CHECK_ARG(a->K > 0);
    if (a->work) {                               // bf16 operand images + DMA-staged kernel (gemm_bf16.hip)
        const int took = FT_OPNAME(ftint_gemm_bf16)(a, reinterpret_cast<hipStream_t>(stream));
        if (took != 0) return took < 0 ? took : FT_OK;
    }
    GemmP p;
    p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.sAm = a->sAm; p.sAk = a->sAk; p.sBk = a->sBk; p.sBn = a->sBn; p.ldc = a->ldc;
    p.bsA = a->bsA; p.bsB = a->bsB; p.bsC = a->bsC;
    p.alpha = a->alpha; p.beta = a->beta; p.act = a->act;
    p.amode = pick_mode(a->A, a->sAm, a->sAk, a->bsA, a->batch);
    p.bmode = pick_mode(a->B, a->sBn, a->sBk, a->bsB, a->batch);
    const bool can_split = (a->flags & FT_GEMM_SPLITK) && a->act == FT_ACT_NONE && (a->beta == 0.f || a->beta == 1.f) &&
                           a->batch == 1 && a->K >= 2048;
    p.gx = cdiv(a->N, BN); p.gy = cdiv(a->M, BM);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // split-K for GEMMs with few output tiles and a long reduction (weight gradients: K = T*B rows): partial
    // products are combined with fp32 global atomics into a zeroed (beta 0) or pre-loaded (beta 1) C.
    p.splits = 1;
    const long tiles = (long)p.gx * p.gy * a->batch;
    if (can_split && tiles < 512) {
        long s = 768 / tiles;
        const long smax = a->K / 512;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
        if (s > 1) p.splits = (int)s;
    }
    p.kchunk = cdiv(cdiv(a->K, BK), p.splits) * BK;
    p.splits = cdiv(a->K, p.kchunk);
    if (p.splits > 1 && a->beta == 0.f)
        FT_CHECK_HIP(hipMemset2DAsync(a->C, sizeof(float) * a->ldc, 0, sizeof(float) * a->N, a->M, st));
    dim3 grid(p.gx * p.gy, p.splits, a->batch);
    FT_CHECK_ARG(grid.z <= 65535);
    if (a->mode == FT_F32) launch_a<0>(p, grid, st);
    else launch_a<1>(p, grid, st);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
