// bf16-image GEMM for the large projections of the training step (FT_BF16 mode, batch == 1, caller-provided workspace).
//
// The fp32-staging kernel of gemm.hip rounds its operands to bf16 on the way into LDS, so every k-step moves 4-byte
// operands through L2 -> VGPR -> cvt -> ds_write: 125 B/clk/CU at the full MFMA rate, twice what the L2 delivers, and the
// transposed operands of the weight gradients pay a register transpose on top.  Here the rounding happens ONCE per
// operand in a streaming pre-pass that also puts the reduction dimension innermost and zero-pads to whole tiles:
//     k contiguous source  -> image [ceil128(rows)][ceil32(K)]      (also generic strides)
//     row contiguous source -> image [ceil32(K)][ceil128(rows)]     (k-major: NO transpose, the kernel reads it with
//                                                                    the LDS transpose-read ds_read_b64_tr_b16)
// and the GEMM proper (templated on the layout of each operand) has no bounds checks in its main loop:
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

inline size_t up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------------------------
// operand images: dst[r][c] = bf16(src(r,c)) for r < R, c < Cc, zero elsewhere; dst is [Rp][Cp] (Cp % 8 == 0)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 o;
    o.x = pack_op16x2(v[0], v[1]); o.y = pack_op16x2(v[2], v[3]);
    o.z = pack_op16x2(v[4], v[5]); o.w = pack_op16x2(v[6], v[7]);
    return o;
}

// Row map (compact images, ft_bf16_image_rows): image row i holds source row rowmap[i] for i < *rdev (-1 = a zero row); rows up
// to ceil256(*rdev + 32) are written (zeros beyond *rdev: the reduction dimension of the weight-gradient GEMMs runs over image
// rows in 32-row steps, and a one-row shift may look one step further), the rest of the worst-case sized buffer is never touched.
__device__ __forceinline__ int mapped_rows(const int* rdev, int Rp, int& Rz) {
    const int R = *rdev;
    const int z = (R + 32 + 255) & ~255;
    Rz = z < Rp ? z : Rp;
    return R;
}

// src(r,c) = src[r*sr + c*sc]; vec: sc == 1, 16-byte aligned rows
__global__ __launch_bounds__(256) void img_rows_k(const float* __restrict__ src, long sr, long sc, int R, int Cc,
                                                  unsigned short* __restrict__ dst, int Rp, int Cp, int vec,
                                                  const int* __restrict__ rowmap, const int* __restrict__ rdev) {
    const int cq = Cp >> 3;
    int Rz = Rp;
    if (rdev) R = mapped_rows(rdev, Rp, Rz);
    const size_t total = (size_t)Rz * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ri = (int)(i / cq), c = (int)(i % cq) * 8;
        int r = ri;
        if (rowmap) r = ri < R ? rowmap[ri] : -1;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r >= 0 && (rowmap || r < R) && c < Cc) {
            const float* p = src + (size_t)r * sr + (size_t)c * sc;
            if (vec && c + 7 < Cc) {
                const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (c + e < Cc) v[e] = p[(size_t)e * sc];
            }
        }
        *reinterpret_cast<uint4*>(dst + (size_t)ri * Cp + c) = pack8(v);
    }
}

// same image, plus the fp32 column sums of the SOURCE (bias gradients: the conversion pass reads the output gradient
// anyway, so ft_colsum's second sweep over it disappears).  Block = 32 column groups x 8 row lanes over a 256-row slab;
// grid (Cp/256, Rp/256); colsum [Cc] must be zero on entry, slabs combine with fp32 atomics (as ft_colsum does).
__global__ __launch_bounds__(256) void img_rows_sum_k(const float* __restrict__ src, long sr, int R, int Cc,
                                                      unsigned short* __restrict__ dst, int Rp, int Cp, int vec,
                                                      float* __restrict__ colsum, const int* __restrict__ rowmap,
                                                      const int* __restrict__ rdev) {
    __shared__ float red[8][32][9];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + tx) * 8;
    const int r0 = blockIdx.y * 256;
    int Rz = Rp;
    if (rdev) R = mapped_rows(rdev, Rp, Rz);
    if (r0 >= Rz) return;                                           // whole slab beyond the mapped rows (uniform)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < Cp) {
#pragma unroll 4
        for (int k = 0; k < 32; ++k) {
            const int ri = r0 + ty + 8 * k;
            if (ri >= Rz) break;
            int r = ri;
            if (rowmap) r = ri < R ? rowmap[ri] : -1;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r >= 0 && (rowmap || r < R) && c < Cc) {
                const float* p = src + (size_t)r * sr + c;
                if (vec && c + 7 < Cc) {
                    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (c + e < Cc) v[e] = p[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
            *reinterpret_cast<uint4*>(dst + (size_t)ri * Cp + c) = pack8(v);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx][e] = acc[e];
    __syncthreads();
    const int cx = threadIdx.x >> 3, e = threadIdx.x & 7;          // 32 column groups x 8 elements = 256 columns
    float s8 = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) s8 += red[y][cx][e];
    const int col = (blockIdx.x * 32 + cx) * 8 + e;
    if (col < Cc) atomicAdd(colsum + col, s8);
}

void make_image(const float* src, long sr, long sc, int R, int Cc, unsigned short* dst, int Rp, int Cp, hipStream_t st,
                const int* rowmap = nullptr, const int* rdev = nullptr) {
    const bool vec = sc == 1 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && sr % 4 == 0;
    const size_t chunks = (size_t)Rp * (Cp >> 3);
    const int blocks = (int)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
    hipLaunchKernelGGL(img_rows_k, dim3(blocks), dim3(256), 0, st, src, sr, sc, R, Cc, dst, Rp, Cp, vec ? 1 : 0, rowmap, rdev);
}

// ---------------------------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------------------------
struct BfP {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, nk;                           // nk = number of 32-wide k-steps
    long lda, ldb, ldc;                     // image row strides in elements, C in floats
    float alpha, beta;
    int act, gx, gy, splits, ksteps;        // tile grid, split-K factor, k-steps per split
    int vec_c;                              // C rows are 16-byte aligned: float4 epilogue
    // compact row space (ft_gemm_img_args.compact): 1 = the M rows are compact rows -- tiles at or beyond *rows_dev exit, C row
    // `rowmap[m]` receives compact row m (negative / beyond *rows_dev: dropped); 2 = the reduction runs over compact rows --
    // k-steps at or beyond *rows_dev - k_shift are not visited (the images are zero there up to the next 32-row step)
    const int* rowmap; const int* rows_dev; int compact, k_shift;
    int chunk_w;                            // > 0: L2-aware tile order with column chunks of this many tiles (no split-K)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

// One operand = a 128 (rows: m or n) x 32 (k) tile per stage, 8 KiB, as eight 1 KiB sub-tiles = eight wave-wide DMAs.
//   KM = false (k contiguous in the image, [row][k]):  sub-tile rg = [16 rows][32 k]; fragment by ds_read_b128.
//   KM = true  (k-major image, [k][row]):              sub-tile (mb, kq) = [8 k][64 rows] (128-byte image rows, so the DMA
//        reads whole 128-byte lines); the MFMA fragment (8 consecutive k of one row per lane) is two ds_read_b64_tr_b16:
//        in a 16-lane group, lane 4j+q supplies the address of [k = 4h+j][rows 4q..4q+3] and lane c receives the four k of
//        row c (probed on gfx950: scripts/exp/tr_probe.hip).  Bank swizzle on the DMA source: the 16-byte piece c of image
//        row (kq, r) lands in slot c ^ 2*s, s = ((r>>1)&1) ^ ((kq&1)<<1), which spreads the four k-rows of a read and the two
//        k-groups of a 32-lane half over distinct 32-byte bank groups.
template <bool KM, int RT = 128>              // RT = rows of the workgroup tile this operand covers (128, or 256 for A)
struct Operand {
    static constexpr int NS = RT / 64;         // sub-tiles (DMAs) per wave and stage
    static constexpr int WT = RT / 32;         // 16-row fragments per wave tile (wave tile = RT/2 rows)
    const unsigned short* src[NS];             // this lane's DMA source for its wave's sub-tiles (k-step 0)
    size_t kstride;                            // elements to advance per k-step
    int dst[NS];                               // byte offsets of those sub-tiles inside the operand's stage buffer
    int roff;                                  // this lane's fragment read offset (without the fragment term)
    __device__ __forceinline__ void init(const unsigned short* img, long ld, int r0, int wave, int lane, int wsel) {
        const int li = lane & 15, kg = lane >> 4;
        if constexpr (!KM) {
            const int srow = lane >> 2, skp = (lane & 3) ^ ((lane >> 4) & 2);
#pragma unroll
            for (int g = 0; g < NS; ++g) {
                src[g] = img + (size_t)(r0 + (wave * NS + g) * 16 + srow) * ld + skp * 8;
                dst[g] = (wave * NS + g) << 10;
            }
            kstride = 32;
            roff = (wsel * WT << 10) + (li * 4 + (kg ^ ((li >> 2) & 2))) * 16;
        } else {
            const int r = lane >> 3, kq = wave;
            const int s = ((lane >> 4) & 1) ^ ((kq & 1) << 1);
            const int c = (lane & 7) ^ (2 * s);
#pragma unroll
            for (int g = 0; g < NS; ++g) {
                src[g] = img + (size_t)(kq * 8 + r) * ld + r0 + g * 64 + c * 8;
                dst[g] = (g * 4 + kq) << 10;
            }
            kstride = (size_t)32 * ld;
            const int sr = ((li >> 3) & 1) ^ ((kg & 1) << 1);
            roff = ((wsel * (WT / 4) * 4 + kg) << 10) + (li >> 2) * 128 + sr * 32 + (li & 3) * 8;
        }
    }
    __device__ __forceinline__ void issue(unsigned char* sbuf, int t) const {
#pragma unroll
        for (int g = 0; g < NS; ++g)
            __builtin_amdgcn_global_load_lds((glb_void*)(src[g] + (size_t)t * kstride), (lds_void*)(sbuf + dst[g]), 16, 0, 0);
    }
    // k-contiguous image: the whole fragment is one ds_read_b128 (compiler-scheduled)
    __device__ __forceinline__ bf16x8 frag(const unsigned char* sbuf, int i) const {
        return *reinterpret_cast<const bf16x8*>(sbuf + roff + (i << 10));
    }
    // k-major image: issue the two transpose-reads of fragment i.  Inline asm on purpose: behind the builtin hipcc waits
    // vmcnt(0) before the first LDS read of every k-step, which drains the DMAs of the NEXT stage that were just issued; the
    // caller retires the reads with tr_wait() (the "+v" operands order every consumer behind the wait).
    __device__ __forceinline__ void tr_issue(const unsigned char* sbuf, int i, bf16x4& lo, bf16x4& hi) const {
        const unsigned int a = (unsigned int)(size_t)(lds_void*)(sbuf + ((roff + ((i >> 2) << 12)) ^ ((i & 3) * 32)));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(a));
    }
};

__device__ __forceinline__ void tr_wait(bf16x4 (&t)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
}
__device__ __forceinline__ void tr_wait(bf16x4 (&t)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                 "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
}
__device__ __forceinline__ void tr_wait(bf16x4 (&t)[16], bf16x4 (&u)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                 "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]),
                 "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
}
__device__ __forceinline__ void tr_wait(bf16x4 (&t)[8], bf16x4 (&u)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                 "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
}

// RTA = rows of the workgroup tile (128, or 256: wave tile 128 x 64 = 8 x 4 MFMA tiles -- 12 fragment reads feed 32 MFMAs instead of
// 8 feeding 16, which takes the LDS pipe off the critical path; 2 workgroups per CU, 196 VGPRs)
template <bool AKM, bool BKM, bool SPLIT, int RTA>
__global__ __launch_bounds__(256, RTA == 256 ? 2 : 4) void gemm_bf16_k(BfP p) {
    constexpr int OPA = RTA * 64, OPB = 8192;          // bytes per operand per stage
    constexpr int TI = RTA / 32;                       // 16-row fragments of the wave tile along M
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (OPA + OPB)];     // [stage][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    const int t0 = blockIdx.y * p.ksteps;
    int t1 = (t0 + p.ksteps < p.nk) ? t0 + p.ksteps : p.nk;
    int rows_lim = p.M, gy = p.gy;
    if (p.compact) {
        const int R = __builtin_amdgcn_readfirstlane(*p.rows_dev);
        if (p.compact == 1) {
            // the grid was sized for the capacity: only the first gx * ceil(R / RTA) workgroups have a tile, and the XCD-aware
            // order below is formed over THOSE (else the XCDs that own the tail of the capacity would sit idle)
            rows_lim = R < p.M ? R : p.M;
            gy = (rows_lim + RTA - 1) / RTA;
            if (gy == 0) return;
        } else {
            const int nk_eff = (R - p.k_shift + 31) >> 5;
            t1 = t1 < nk_eff ? t1 : nk_eff;
            if (SPLIT && t0 >= t1) return;                          // this k-slice lies entirely in the unused tail
        }
    }
    int mt, nt;
    if (p.chunk_w > 0 && gy >= 16) {
        // L2-aware order (workgroup L runs on XCD L % 8, in launch order).  An XCD owns a contiguous range of tile ROWS and walks it
        // in column chunks of chunk_w tiles, row-major inside a chunk: the B panels of the chunk (<= ~2 MB, chosen by the host
        // from K) stay in that XCD's 4 MiB L2 while every A panel of the range is streamed past them once, and the chunk_w
        // workgroups that share an A panel run back to back.  The plain order below keeps 32 B panels (the whole 13.6 MB weight
        // image of an LSTM input projection) live per XCD: every tile then re-reads its B panel from beyond the L2.
        const int xcd = tile & 7, idx = tile >> 3;
        const int qm = gy >> 3, rm = gy & 7;
        const int mh = qm + (xcd < rm ? 1 : 0);
        const int m_lo = xcd * qm + (xcd < rm ? xcd : rm);
        if (idx >= mh * p.gx) return;
        const int per_chunk = mh * p.chunk_w;
        const int ch = idx / per_chunk, within = idx - ch * per_chunk;
        const int left = p.gx - ch * p.chunk_w;
        const int cw = left < p.chunk_w ? left : p.chunk_w;
        const int mi = within / cw;
        mt = m_lo + mi;
        nt = ch * p.chunk_w + (within - mi * cw);
    } else {
        // XCD-aware order: every XCD gets a contiguous run of tiles, x fastest
        const int total = p.gx * gy, q = total >> 3, r = total & 7;
        const int xcd = tile & 7, idx = tile >> 3;
        if (tile >= total) return;
        tile = xcd * q + (xcd < r ? xcd : r) + idx;
        mt = tile / p.gx;
        nt = tile % p.gx;
    }
    const int m0 = mt * RTA, n0 = nt * TB;

    Operand<AKM, RTA> oa;
    Operand<BKM, 128> ob;
    oa.init(p.A, p.lda, m0, wave, lane, wm);
    ob.init(p.B, p.ldb, n0, wave, lane, wn);

    // !SPLIT: acc[i][j] holds C^T: lane (li, kg), register r  <->  C[m = i*16 + li][n = j*16 + kg*4 + r]  (operands swapped
    //         in the MFMA so that a lane owns four CONSECUTIVE output columns: float4 stores / loads in the epilogue)
    //  SPLIT: natural order, register r <-> C[m = i*16 + kg*4 + r][n = j*16 + li]: one atomic instruction then covers 16
    //         consecutive columns of 4 rows (4 cache lines) instead of 4 columns of 16 rows (measured 20-60 % faster)
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (t0 < t1) { oa.issue(smem, t0); ob.issue(smem + OPA, t0); }
    for (int t = t0; t < t1; ++t) {
        const int stage = (t - t0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my DMAs of step t have landed
        __syncthreads();                                            // ... everyone's have; stage^1 is no longer being read
        if (t + 1 < t1) {
            unsigned char* nx = smem + (stage ^ 1) * (OPA + OPB);
            oa.issue(nx, t + 1);
            ob.issue(nx + OPA, t + 1);
        }
        const unsigned char* sa = smem + stage * (OPA + OPB);
        const unsigned char* sb = sa + OPA;
        bf16x8 a[TI], b[4];
        bf16x4 ta[2 * TI], tb[8];
        if constexpr (!AKM) {
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = oa.frag(sa, i);
        }
        if constexpr (!BKM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = ob.frag(sb, j);
        }
        if constexpr (AKM) {
#pragma unroll
            for (int i = 0; i < TI; ++i) oa.tr_issue(sa, i, ta[2 * i], ta[2 * i + 1]);
        }
        if constexpr (BKM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ob.tr_issue(sb, j, tb[2 * j], tb[2 * j + 1]);
        }
        if constexpr (AKM && BKM) tr_wait(ta, tb);
        else if constexpr (AKM) tr_wait(ta);
        else if constexpr (BKM) tr_wait(tb);
        if constexpr (AKM) {
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = __builtin_shufflevector(ta[2 * i], ta[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        if constexpr (BKM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = __builtin_shufflevector(tb[2 * j], tb[2 * j + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SPLIT) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
                else acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
            }
    }

    if constexpr (SPLIT) {          // C was zeroed (beta == 0) or holds the addend (beta == 1)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + wm * (RTA / 2) + i * 16 + kg * 4 + r;
                if (row >= rows_lim) continue;
                if (p.compact == 1) { row = p.rowmap[row]; if (row < 0) continue; }
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
    for (int i = 0; i < TI; ++i) {
        int row = m0 + wm * (RTA / 2) + i * 16 + li;
        if (row >= rows_lim) continue;
        if (p.compact == 1) { row = p.rowmap[row]; if (row < 0) continue; }
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

template <bool AKM, bool BKM>
void launch_s(const BfP& p, dim3 grid, bool big, hipStream_t st) {
    if (big) {
        if (p.splits > 1) hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, true, 256>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, false, 256>), grid, dim3(256), 0, st, p);
    } else {
        if (p.splits > 1) hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, true, 128>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, false, 128>), grid, dim3(256), 0, st, p);
    }
}

// images -> C.  a_km / b_km: the operand image is k-major ([k][row]) instead of k-contiguous ([row][k]).
// Images are padded to multiples of 256 in both dimensions (ft_bf16_image), so either tile height may run off the logical M.
int run_images(const unsigned short* A, long lda, int a_km, const unsigned short* B, long ldb, int b_km, float* C, long ldc,
               const float* bias, int M, int N, int K, float alpha, float beta, int act, int flags, hipStream_t st,
               const int* rowmap = nullptr, const int* rows_dev = nullptr, int compact = 0, int k_shift = 0) {
    static const int force_tile = [] { const char* e = getenv("FT_GEMM_BF16_TILE"); return e ? atoi(e) : 0; }();
    BfP p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.nk = cdiv(K, 32); p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta; p.act = act;
    p.rowmap = rowmap; p.rows_dev = rows_dev; p.compact = compact; p.k_shift = k_shift;
    const bool can_split = (flags & FT_GEMM_SPLITK) && act == FT_ACT_NONE && (beta == 0.f || beta == 1.f) && K >= 2048;
    // 256 x 128 workgroup tiles (2 per CU) when they still fill the chip, possibly with split-K; else 128 x 128 (4 per CU)
    const long tiles_big = (long)cdiv(M, 256) * cdiv(N, TB);
    // measured (scripts/exp/gemm_bench.py): the tall tile pays for the long-K weight-gradient shapes (+9 %), is neutral to
    // slightly slower for the forward / input-gradient shapes -- those keep the 128 x 128 tile at 4 workgroups per CU
    bool big = M >= 512 && can_split && tiles_big * (K / 512) >= 384;
    if (force_tile == 128) big = false;
    if (force_tile == 256) big = M >= 256;
    const int RTA = big ? 256 : TB;
    const long slots = big ? 512 : 1024;
    p.gx = cdiv(N, TB); p.gy = cdiv(M, RTA);
    p.vec_c = (reinterpret_cast<uintptr_t>(C) % 16 == 0 && ldc % 4 == 0) ? 1 : 0;
    const long tiles = (long)p.gx * p.gy;
    long s = 1;
    if (can_split && tiles < slots / 2) {
        s = slots / tiles;                     // fill all workgroup slots of the 256 CUs
        const long smax = K / 512;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    if (compact == 1) s = 1;                   // scattered output rows: no split-K (the zero fill would have to be scattered too)
    // compact reduction: K is the capacity, ~30 % of the k-slices of a typical batch are empty -- over-split to keep the slots filled
    if (compact == 2 && s > 1) { s = s * 4 / 3; if (s > K / 512) s = K / 512; if (s > 64) s = 64; if (s < 1) s = 1; }
    p.ksteps = cdiv(p.nk, s);
    p.splits = cdiv(p.nk, p.ksteps);
    if (p.splits > 1 && beta == 0.f) FT_CHECK_HIP(hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, st));
    // L2-aware tile order for the un-split kernels: column chunks whose B panels (chunk_w x TB rows x K) fit ~2 MB of an XCD's L2
    static const int order_on = [] { const char* e = getenv("FT_GEMM_BF16_ORDER"); return e ? atoi(e) : 1; }();
    p.chunk_w = 0;
    int gridx = p.gx * p.gy;
    if (order_on && p.splits == 1 && p.gy >= 16) {
        long cw = (2l << 20) / ((long)TB * (long)p.nk * 32 * 2);
        p.chunk_w = (int)(cw < 1 ? 1 : (cw > p.gx ? p.gx : cw));
        gridx = 8 * ((p.gy + 7) / 8) * p.gx;          // every XCD is handed the blocks of the largest row range
    }
    const dim3 grid(gridx, p.splits);
    if (a_km) { if (b_km) launch_s<true, true>(p, grid, big, st); else launch_s<true, false>(p, grid, big, st); }
    else      { if (b_km) launch_s<false, true>(p, grid, big, st); else launch_s<false, false>(p, grid, big, st); }
    FT_CHECK_LAUNCH();
    return FT_OK;
}

bool qualifies(const ft_gemm_args* a) {
    return a->mode == FT_OP16 && a->batch == 1 && a->M >= 32 && a->N >= 32 && a->K >= 16 &&
           (double)a->M * a->N * a->K >= (double)(1 << 20);
}

// image geometry of one ft_gemm operand: rows index `R` (m or n) with stride sr, reduction stride sk
struct ImgGeo { int km; int rows, cols; long sr, sc; int Rp, Cp; size_t bytes; };
ImgGeo geo(long sr, long sk, int R, int K) {
    ImgGeo g;
    g.km = (sr == 1 && sk != 1) ? 1 : 0;               // row dim contiguous in the source: keep it k-major, no transpose
    if (g.km) { g.rows = K; g.cols = R; g.sr = sk; g.sc = 1; g.Rp = (int)up(K, 32); g.Cp = (int)up(R, 256); }
    else      { g.rows = R; g.cols = K; g.sr = sr; g.sc = sk; g.Rp = (int)up(R, 256); g.Cp = (int)up(K, 32); }
    g.bytes = up((size_t)g.Rp * g.Cp * 2, 256);
    return g;
}

}  // namespace

#if FT_OPFMT == 0
extern "C" size_t ft_gemm_workspace_bytes(const ft_gemm_args* a) {          // format-independent: both modes use 2-byte images
    if (!a) return 0;
    ft_gemm_args b = *a;
    if (b.mode == FT_F16) b.mode = FT_BF16;
    if (!qualifies(&b)) return 0;
    return geo(a->sAm, a->sAk, a->M, a->K).bytes + geo(a->sBn, a->sBk, a->N, a->K).bytes;
}
#endif

// returns 1 when the call was taken by this path, 0 when the caller should use the fp32-staging kernel, < 0 on error
int FT_OPNAME(ftint_gemm_bf16)(const ft_gemm_args* a, hipStream_t st) {
    if (!qualifies(a) || !a->work) return 0;
    const ImgGeo ga = geo(a->sAm, a->sAk, a->M, a->K), gb = geo(a->sBn, a->sBk, a->N, a->K);
    if (a->work_bytes < ga.bytes + gb.bytes) return 0;
    if (reinterpret_cast<uintptr_t>(a->work) % 256 != 0) return ft_fail(FT_EINVAL, "ft_gemm: work must be 256-byte aligned");
    unsigned short* Aimg = reinterpret_cast<unsigned short*>(a->work);
    unsigned short* Bimg = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(a->work) + ga.bytes);
    make_image(a->A, ga.sr, ga.sc, ga.rows, ga.cols, Aimg, ga.Rp, ga.Cp, st);
    make_image(a->B, gb.sr, gb.sc, gb.rows, gb.cols, Bimg, gb.Rp, gb.Cp, st);
    const int rc = run_images(Aimg, ga.Cp, ga.km, Bimg, gb.Cp, gb.km, a->C, a->ldc, a->bias, a->M, a->N, a->K, a->alpha, a->beta,
                              a->act, a->flags, st);
    return rc < 0 ? rc : 1;
}

// ---------------------------------------------------------------------------------------------------------------
// explicit images: the caller converts each fp32 matrix ONCE and feeds it to every GEMM that reads it -- an activation to
// its forward GEMM and its weight-gradient GEMM, an output gradient to the input-gradient and the weight-gradient GEMM, a
// weight to forward and backward -- in whichever role (k-contiguous or k-major) that GEMM needs.
// ---------------------------------------------------------------------------------------------------------------
#if FT_OPFMT == 0
extern "C" size_t ft_bf16_image_bytes(int64_t rows, int64_t cols) {
    if (rows < 1 || cols < 1) return 0;
    return up(up((size_t)rows + 32, 256) * up((size_t)cols, 256) * 2, 256);
}
#endif

extern "C" int FT_OPNAME(ft_bf16_image)(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, void* stream) {
    FT_CHECK_ARG(src && dst && rows >= 1 && cols >= 1 && ld >= cols && rows < (1ll << 31) - 256 && cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    make_image(src, ld, 1, (int)rows, (int)cols, reinterpret_cast<unsigned short*>(dst), (int)up((size_t)rows + 32, 256),
               (int)up((size_t)cols, 256), reinterpret_cast<hipStream_t>(stream));
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_bf16_image_colsum)(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream) {
    FT_CHECK_ARG(src && dst && colsum && rows >= 1 && cols >= 1 && ld >= cols && rows < (1ll << 31) - 256 && cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Rp = (int)up((size_t)rows + 32, 256), Cp = (int)up((size_t)cols, 256);
    const int vec = (reinterpret_cast<uintptr_t>(src) % 16 == 0 && ld % 4 == 0) ? 1 : 0;
    FT_CHECK_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)cols, st));
    hipLaunchKernelGGL(img_rows_sum_k, dim3(cdiv(Cp, 256), cdiv(Rp, 256)), dim3(256), 0, st, src, (long)ld, (int)rows, (int)cols,
                       reinterpret_cast<unsigned short*>(dst), Rp, Cp, vec, colsum, nullptr, nullptr);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_gemm_img)(const ft_gemm_img_args* a, void* stream) {
    FT_CHECK_ARG(a != nullptr);
    FT_CHECK_ARG(a->A && a->B && a->C && a->M >= 1 && a->N >= 1 && a->K >= 1);
    FT_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && reinterpret_cast<uintptr_t>(a->A) % 16 == 0 && reinterpret_cast<uintptr_t>(a->B) % 16 == 0);
    FT_CHECK_ARG(a->compact >= 0 && a->compact <= 2 && (a->compact == 0 || a->rows_dev) && (a->compact != 1 || a->rowmap));
    FT_CHECK_ARG(a->k_shift >= 0 && (a->compact == 2 || a->k_shift == 0));
    return run_images(reinterpret_cast<const unsigned short*>(a->A), a->lda, a->a_kmajor, reinterpret_cast<const unsigned short*>(a->B),
                      a->ldb, a->b_kmajor, a->C, a->ldc, a->bias, a->M, a->N, a->K, a->alpha, a->beta, a->act, a->flags,
                      reinterpret_cast<hipStream_t>(stream), a->rowmap, a->rows_dev, a->compact, a->k_shift);
}

// compact image: image row i = source row rowmap[i] (i < *rows_dev; -1 = zero row); buffer sized for cap_rows
extern "C" int FT_OPNAME(ft_bf16_image_rows)(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                                             const int32_t* rowmap, const int32_t* rows_dev, void* stream) {
    FT_CHECK_ARG(src && dst && rowmap && rows_dev && cap_rows >= 1 && cols >= 1 && ld >= cols && cap_rows < (1ll << 31) - 512 && cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Rp = (int)up((size_t)cap_rows + 32, 256), Cp = (int)up((size_t)cols, 256);
    if (!colsum) {
        make_image(src, ld, 1, (int)cap_rows, (int)cols, reinterpret_cast<unsigned short*>(dst), Rp, Cp, st, rowmap, rows_dev);
    } else {
        const int vec = (reinterpret_cast<uintptr_t>(src) % 16 == 0 && ld % 4 == 0) ? 1 : 0;
        FT_CHECK_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)cols, st));
        hipLaunchKernelGGL(img_rows_sum_k, dim3(cdiv(Cp, 256), cdiv(Rp, 256)), dim3(256), 0, st, src, (long)ld, (int)cap_rows, (int)cols,
                           reinterpret_cast<unsigned short*>(dst), Rp, Cp, vec, colsum, rowmap, rows_dev);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
