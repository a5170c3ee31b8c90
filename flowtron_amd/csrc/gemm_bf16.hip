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
//   * 128x128 tile (256x128 for the split-K weight gradients), 4 waves (2x2, 64x64 per wave = 4x4 MFMA 16x16x32 tiles per k-step),
//   * operands go global -> LDS by `global_load_lds_dwordx4` (no VGPR staging, no ds_write pass), one barrier pair per stage,
//     4 workgroups per CU (32 KiB LDS each),
//   * store kernels with a k-contiguous operand (forward, input gradients; round 6): 64-wide k stages in ONE buffer -- a DMA piece is
//     8 image rows x 128 B (whole cache lines), the LDS image has 128-byte rows with the 16-byte chunk c of row r at c ^ ((r >> 1) & 7);
//     the DMAs of the next stage go out once every wave holds the stage's last fragments in registers (Operand64, template SB),
//   * 32-wide double-buffered stages elsewhere (both operands k-major: the weight gradients; K not a multiple of 64): the LDS image is a
//     sequence of 1 KiB [16 rows][32 k] sub-tiles = exactly one wave-wide DMA each; the DMA writes lane-linear, so the bank swizzle is
//     applied to the SOURCE address: LDS slot(row, kpart) = row*4 + (kpart ^ ((row>>2)&2)), which makes the four 16-lane service groups
//     of ds_read_b128 ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte slots.
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

// src(r,c) = src[r*sr + c*sc]; vec: sc == 1, 16-byte aligned rows.  (first, stride): this thread's first chunk and the launch's
// chunk stride -- the whole grid for img_rows_k, a block range of the descriptor-table launch for img_table_k
__device__ __forceinline__ void img_rows_body(const float* __restrict__ src, long sr, long sc, int R, int Cc,
                                              unsigned short* __restrict__ dst, int Rp, int Cp, int vec,
                                              const int* __restrict__ rowmap, const int* __restrict__ rdev, long dld,
                                              size_t first, size_t stride) {
    const int cq = Cp >> 3;
    int Rz = Rp;
    if (rdev) R = mapped_rows(rdev, Rp, Rz);
    const size_t total = (size_t)Rz * cq;
    for (size_t i = first; i < total; i += stride) {
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
        *reinterpret_cast<uint4*>(dst + (size_t)ri * dld + c) = pack8(v);
    }
}
__global__ __launch_bounds__(256) void img_rows_k(const float* __restrict__ src, long sr, long sc, int R, int Cc,
                                                  unsigned short* __restrict__ dst, int Rp, int Cp, int vec,
                                                  const int* __restrict__ rowmap, const int* __restrict__ rdev, long dld) {
    img_rows_body(src, sr, sc, R, Cc, dst, Rp, Cp, vec, rowmap, rdev, dld, blockIdx.x * (size_t)blockDim.x + threadIdx.x,
                  (size_t)gridDim.x * blockDim.x);
}

// Split image (ft_bf16_image_split3): x = hi + lo with hi = op16(x), lo = op16(x - hi) (the pair carries ~16 significand bits).  The
// image is three column blocks of K: an ACTIVATION as [hi | lo | hi], a WEIGHT as [hi | hi | lo], so that ONE 16-bit GEMM over
// 3 K columns yields x_hi w_hi + x_lo w_hi + x_hi w_lo = x w up to the dropped lo . lo term (2^-18): fp32-grade products at three
// times a 16-bit GEMM's cost instead of the fp32 MFMA's sixteen.  Rows [R, Rp) and columns [3 K, Cp) are zero.  K % 8 == 0.
// (Both correction terms are needed: with either one left out the encoder's worst gradient deviation reads 0.096 instead of 0.008.)
__device__ __forceinline__ void img_split3_body(const float* __restrict__ src, long sr, int R, int K, unsigned short* __restrict__ dst,
                                                int Rp, int Cp, int weight, size_t first, size_t stride) {
    const int kq = K >> 3, cq = Cp >> 3;
    const size_t total = (size_t)Rp * cq;
    for (size_t i = first; i < total; i += stride) {
        const int r = (int)(i / cq), q = (int)(i % cq);
        uint4 out = make_uint4(0u, 0u, 0u, 0u);
        if (r < R && q < 3 * kq) {
            const int blk = q / kq, c = (q - blk * kq) * 8;
            const float* p = src + (size_t)r * sr + c;
            float v[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p[e];
            const bool want_lo = weight ? blk == 2 : blk == 1;
            if (want_lo) {
#pragma unroll
                for (int e = 0; e < 8; ++e) lo[e] = v[e] - op16_to_f(f2op16(v[e]));
                out = pack8(lo);
            } else {
                out = pack8(v);
            }
        }
        *reinterpret_cast<uint4*>(dst + (size_t)r * Cp + (size_t)q * 8) = out;
    }
}
__global__ __launch_bounds__(256) void img_split3_k(const float* __restrict__ src, long sr, int R, int K, unsigned short* __restrict__ dst,
                                                    int Rp, int Cp, int weight) {
    img_split3_body(src, sr, R, K, dst, Rp, Cp, weight, blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// The split ACTIVATION image [hi | lo | hi] of the encoder convolution's im2col matrix, taken straight from x (ft_bf16_image_split3_im2col):
// image row r = l * B + b, column j = c * KW + k holds x[l + k - KW / 2][b][c] where that position exists in the utterance (ft_im2col's
// rule), so the fp32 [L * B][C * KW] matrix -- 51 MB per layer at the bench shape, written once and read once -- never exists.
__global__ __launch_bounds__(256) void img_split3_im2col_k(const float* __restrict__ x, const int* __restrict__ lens, int Lx, int B, int Cc, int KW,
                                                           unsigned short* __restrict__ dst, int Rp, int Cp) {
    const int R = Lx * B, K = Cc * KW;
    const int kq = K >> 3, cq = Cp >> 3;
    const size_t total = (size_t)Rp * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cq), q = (int)(i % cq);
        uint4 out = make_uint4(0u, 0u, 0u, 0u);
        if (r < R && q < 3 * kq) {
            const int blk = q / kq, c0 = (q - blk * kq) * 8;
            const int l = r / B, b = r - l * B;
            const int len = lens[b];
            float v[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = c0 + e, ch = j / KW, k = j - ch * KW;
                const int ls = l + k - KW / 2;
                v[e] = (ls >= 0 && ls < len) ? x[((size_t)ls * B + b) * Cc + ch] : 0.f;
            }
            if (blk == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) lo[e] = v[e] - op16_to_f(f2op16(v[e]));
                out = pack8(lo);
            } else {
                out = pack8(v);
            }
        }
        *reinterpret_cast<uint4*>(dst + (size_t)r * Cp + (size_t)q * 8) = out;
    }
}

// Descriptor table (ft_bf16_image_table): the plain and split WEIGHT images of one forward pass in ONE launch -- a training step
// rounds 23 weight matrices (60 M parameters) afresh, each in a launch of 5-15 us for 1-5 us of work.  Workgroup ranges per image
// (blk0), each range walks its image with the single-image kernels' own bodies.
constexpr int IMG_TABLE_MAX = 32;
struct ImgTable {
    const float* src[IMG_TABLE_MAX];
    unsigned short* dst[IMG_TABLE_MAX];
    long sr[IMG_TABLE_MAX];
    int R[IMG_TABLE_MAX], Cc[IMG_TABLE_MAX], Rp[IMG_TABLE_MAX], Cp[IMG_TABLE_MAX], kind[IMG_TABLE_MAX], vec[IMG_TABLE_MAX];
    int blk0[IMG_TABLE_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void img_table_k(ImgTable t) {
    int d = 0;
    while (d + 1 < t.n && (int)blockIdx.x >= t.blk0[d + 1]) ++d;               // (uniform)
    const size_t first = (size_t)((int)blockIdx.x - t.blk0[d]) * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)(t.blk0[d + 1] - t.blk0[d]) * blockDim.x;
    if (t.kind[d] == 0) img_rows_body(t.src[d], t.sr[d], 1, t.R[d], t.Cc[d], t.dst[d], t.Rp[d], t.Cp[d], t.vec[d], nullptr, nullptr, (long)t.Cp[d], first, stride);
    else img_split3_body(t.src[d], t.sr[d], t.R[d], t.Cc[d], t.dst[d], t.Rp[d], t.Cp[d], 1, first, stride);
}

// same image, plus the fp32 column sums of the SOURCE (bias gradients: the conversion pass reads the output gradient
// anyway, so ft_colsum's second sweep over it disappears).  Block = 32 column groups x 8 row lanes over a 64-row slab;
// grid (Cp/256, Rp/64); colsum [Cc] must be zero on entry, slabs combine with fp32 atomics (as ft_colsum does).
// ysrc != nullptr: src is an OUTPUT GRADIENT dy and ysrc the saved output y = act(pre) of the same rows: the pass images (and sums)
// dpre = dy act'(pre) (ft_act_bwd's formulas) -- the activation backward of a dense layer rides on the conversion pass, no fp32
// dpre tensor is written or read back.
constexpr int IMG_SUM_SLAB = 64;
__global__ __launch_bounds__(256) void img_rows_sum_k(const float* __restrict__ src, long sr, int R, int Cc,
                                                      unsigned short* __restrict__ dst, int Rp, int Cp, int vec,
                                                      float* __restrict__ colsum, const int* __restrict__ rowmap,
                                                      const int* __restrict__ rdev, const float* __restrict__ ysrc = nullptr,
                                                      long ysr = 0, int act = 0) {
    __shared__ float red[8][32][9];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + tx) * 8;
    // (slabs of 64 rows since round 6: a 256-row slab gave a [19 000 x 1024] gradient 300 workgroups -- 1.2 per CU, latency-bound at
    //  1.7 TB/s; four times the workgroups, four times the column-sum atomics: one per column and slab)
    const int r0 = blockIdx.y * IMG_SUM_SLAB;
    int Rz = Rp;
    if (rdev) R = mapped_rows(rdev, Rp, Rz);
    if (r0 >= Rz) return;                                           // whole slab beyond the mapped rows (uniform)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < Cp) {
#pragma unroll 4
        for (int k = 0; k < IMG_SUM_SLAB / 8; ++k) {
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
                if (ysrc) {
                    const float* py = ysrc + (size_t)r * ysr + c;
                    float yv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (vec && c + 7 < Cc) {
                        const float4 a = *reinterpret_cast<const float4*>(py), b = *reinterpret_cast<const float4*>(py + 4);
                        yv[0] = a.x; yv[1] = a.y; yv[2] = a.z; yv[3] = a.w; yv[4] = b.x; yv[5] = b.y; yv[6] = b.z; yv[7] = b.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (c + e < Cc) yv[e] = py[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (act == FT_ACT_TANH) v[e] = v[e] * (1.f - yv[e] * yv[e]);
                        else if (act == FT_ACT_RELU) v[e] = yv[e] > 0.f ? v[e] : 0.f;
                        else if (act == FT_ACT_SIGMOID) v[e] = v[e] * yv[e] * (1.f - yv[e]);
                    }
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

// dld: row stride of the destination (default Cp); a piece of a wider image = dst pointing at its first column, dld the image's stride
void make_image(const float* src, long sr, long sc, int R, int Cc, unsigned short* dst, int Rp, int Cp, hipStream_t st,
                const int* rowmap = nullptr, const int* rdev = nullptr, long dld = 0) {
    const bool vec = sc == 1 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && sr % 4 == 0;
    const size_t chunks = (size_t)Rp * (Cp >> 3);
    const int blocks = (int)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
    hipLaunchKernelGGL(img_rows_k, dim3(blocks), dim3(256), 0, st, src, sr, sc, R, Cc, dst, Rp, Cp, vec ? 1 : 0, rowmap, rdev,
                       dld ? dld : (long)Cp);
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
    // rank-1 epilogue term (ft_gemm_img_args.r1_row / r1_col): C[row][col] += r1row[row] * r1col[col], row = the OUTPUT row
    const float* r1row; const float* r1col;
    // deterministic split-K (FT_GEMM_SPLITK_DET): slice blockIdx.y of the reduction writes ITS partial product, with plain stores, to
    // C + blockIdx.y * c_slice (a workspace); splitk_reduce_k adds the slices in a fixed order.  0: one slice, C is the output
    long c_slice;
    // FT_GEMM_C16: C is a 16-BIT matrix of this build's operand format (ldc in 16-bit elements): the fp32 result (+ bias) is rounded
    // once in the epilogue -- the gx rows a persistent recurrence reads (half the bytes written here and read there)
    int c16;
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

// k-contiguous operand with 64-WIDE k stages: a DMA piece is 8 image rows x 128 B -- whole cache lines, where the 32-wide stage
// above asks the L2 for 16 half lines per piece.  Measured with nothing but the DMAs in the loop (scripts/exp/dma_stream_probe.hip,
// profiles/r06_dma_stream_probe.log): 16-row x 64 B pieces stream 16.5 TB/s into the LDS of the 256 CUs at 4 workgroups per CU,
// 8-row x 128 B pieces 21 TB/s at 2 per CU (24.5 at 4).  LDS image: 128-byte rows, the 16-byte chunk c of row r stored at
// c ^ ((r >> 1) & 7) (swizzle on the DMA source, as gemm_bf16_p256_k); the fragment of k-step ks is one ds_read_b128.
template <int RT>
struct Operand64 {
    static constexpr int NS = RT / 32;         // pieces per wave and stage (the wave stages rows 8 NS wave .. + 8 NS - 1)
    static constexpr int WT = RT / 32;         // 16-row fragments per wave tile
    const unsigned short* src0;                // this lane's source of piece 0 (k = 0, chunk swizzle of the even pieces)
    long ld8;                                  // elements between pieces (8 rows)
    int dst0;                                  // byte offset of piece 0 inside the operand's stage buffer
    int xodd;                                  // element offset that turns the even pieces' chunk into the odd pieces' (c ^ 4)
    int roff;
    __device__ __forceinline__ void init(const unsigned short* img, long ld, int r0, int wave, int lane, int wsel) {
        const int li = lane & 15, kg = lane >> 4;
        const int prow = lane >> 3, pch = lane & 7;
        const int c = pch ^ (prow >> 1);                           // rows 8 (wave NS + g) + prow: ((r >> 1) & 7) = (prow >> 1) | 4 (g & 1)
        src0 = img + (size_t)(r0 + wave * NS * 8 + prow) * ld + c * 8;
        xodd = ((c ^ 4) - c) * 8;
        ld8 = 8 * ld;
        dst0 = wave * NS * 1024;
        roff = (wsel * (RT / 2) + li) * 128 + ((kg ^ (li >> 1)) << 4);
    }
    __device__ __forceinline__ void issue(unsigned char* sbuf, int t) const {       // t counts 64-wide stages
#pragma unroll
        for (int g = 0; g < NS; ++g)
            __builtin_amdgcn_global_load_lds((glb_void*)(src0 + (size_t)t * 64 + (size_t)g * ld8 + ((g & 1) ? xodd : 0)),
                                             (lds_void*)(sbuf + dst0 + g * 1024), 16, 0, 0);
    }
    __device__ __forceinline__ bf16x8 frag(const unsigned char* sbuf, int i, int ks) const {
        return *reinterpret_cast<const bf16x8*>(sbuf + ((roff + (i << 11)) ^ (ks << 6)));
    }
};

// RTA = rows of the workgroup tile (128, or 256: wave tile 128 x 64 = 8 x 4 MFMA tiles -- 12 fragment reads feed 32 MFMAs instead of
// 8 feeding 16, which takes the LDS pipe off the critical path; 2 workgroups per CU, 196 VGPRs)
// KS = 64: 64-wide k stages (two MFMA k-steps per stage; the k-contiguous operands as Operand64, a k-major operand as two of its
// 32-wide sub-stages side by side); t0 / t1 / ksteps keep counting 32-wide steps (t0 even).  SB: a single 32 KiB LDS buffer, 4
// workgroups per CU (with two buffers the stage is 64 KiB and only 2 workgroups fit a CU: slower than 32-wide stages at K = 1024)
template <bool AKM, bool BKM, bool SPLIT, int RTA, int KS = 32, bool SB = false>
__global__ __launch_bounds__(256, SB ? 4 : (RTA == 256 || KS == 64) ? 2 : 4) void gemm_bf16_k(BfP p) {
    static_assert(!SB || (KS == 64 && RTA == 128 && !SPLIT), "single-buffer form: 128 x 128 x 64 store kernels");
    constexpr int KQ = KS / 32;                        // MFMA k-steps per stage
    constexpr int OPA = RTA * 64 * KQ, OPB = 8192 * KQ;     // bytes per operand per stage
    constexpr int TI = RTA / 32;                       // 16-row fragments of the wave tile along M
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(SB ? 1 : 2) * (OPA + OPB)];     // [stage][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    int t0 = blockIdx.y * p.ksteps;
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
            int nk_eff = (R - p.k_shift + 31) >> 5;
            nk_eff = nk_eff < p.nk ? nk_eff : p.nk;
            if constexpr (SPLIT) {
                // the k-slices divide the rows the batch HAS, not the capacity the host sized the grid for: every workgroup of
                // the launch gets the same share of the reduction (the host's even split of the capacity left ~30 % of the slices
                // empty and one partial -- the workgroups of a tile then finished a slice apart)
                const int ks = (nk_eff + (int)gridDim.y - 1) / (int)gridDim.y;
                t0 = blockIdx.y * ks;
                t1 = t0 + ks < nk_eff ? t0 + ks : nk_eff;
                if (t0 >= t1) return;
            } else t1 = t1 < nk_eff ? t1 : nk_eff;
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
    Operand64<RTA> oa64;
    Operand64<128> ob64;
    if constexpr (KS == 64 && !AKM) oa64.init(p.A, p.lda, m0, wave, lane, wm); else oa.init(p.A, p.lda, m0, wave, lane, wm);
    if constexpr (KS == 64 && !BKM) ob64.init(p.B, p.ldb, n0, wave, lane, wn); else ob.init(p.B, p.ldb, n0, wave, lane, wn);

    // !SPLIT: acc[i][j] holds C^T: lane (li, kg), register r  <->  C[m = i*16 + li][n = j*16 + kg*4 + r]  (operands swapped
    //         in the MFMA so that a lane owns four CONSECUTIVE output columns: float4 stores / loads in the epilogue)
    //  SPLIT: natural order, register r <-> C[m = i*16 + kg*4 + r][n = j*16 + li]: one atomic instruction then covers 16
    //         consecutive columns of 4 rows (4 cache lines) instead of 4 columns of 16 rows (measured 20-60 % faster)
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // stage s (of KS k) into the buffer at sbuf
    auto issue_stage = [&](unsigned char* sbuf, int s) {
        if constexpr (KS == 64) {
            if constexpr (!AKM) oa64.issue(sbuf, s); else { oa.issue(sbuf, 2 * s); oa.issue(sbuf + OPA / 2, 2 * s + 1); }
            if constexpr (!BKM) ob64.issue(sbuf + OPA, s); else { ob.issue(sbuf + OPA, 2 * s); ob.issue(sbuf + OPA + OPB / 2, 2 * s + 1); }
        } else { oa.issue(sbuf, s); ob.issue(sbuf + OPA, s); }
    };
    // the fragments of MFMA k-step ks of the stage at sbase, and its TI x 4 MFMAs
    auto load_frags = [&](const unsigned char* sbase, int ks, bf16x8 (&a)[TI], bf16x8 (&b)[4]) {
        const unsigned char* sa = sbase + ((KS == 64 && AKM) ? ks * (OPA / 2) : 0);
        const unsigned char* sb = sbase + OPA + ((KS == 64 && BKM) ? ks * (OPB / 2) : 0);
        bf16x4 ta[2 * TI], tb[8];
        if constexpr (!AKM) {
#pragma unroll
            for (int i = 0; i < TI; ++i) { if constexpr (KS == 64) a[i] = oa64.frag(sa, i, ks); else a[i] = oa.frag(sa, i); }
        }
        if constexpr (!BKM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { if constexpr (KS == 64) b[j] = ob64.frag(sb, j, ks); else b[j] = ob.frag(sb, j); }
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
    };
    auto multiply = [&](const bf16x8 (&a)[TI], const bf16x8 (&b)[4]) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SPLIT) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
                else acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
            }
    };
    const int s0 = t0 / KQ, s1 = (t1 + KQ - 1) / KQ;       // stages of this k-slice
    if (s0 < s1) issue_stage(smem, s0);
    if constexpr (SB) {
        // ONE LDS buffer: the DMAs of stage t + 1 go out as soon as every wave holds the last fragments of stage t in registers, and
        // fly under the second half of its MFMAs -- and under the other three workgroups of the CU, which is where the rest of the
        // round trip hides
        for (int t = s0; t < s1; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                            // stage t has landed for everyone
            bf16x8 a0[TI], b0[4], a1[TI], b1[4];
            load_frags(smem, 0, a0, b0);
            multiply(a0, b0);
            load_frags(smem, 1, a1, b1);
            __syncthreads();                                            // (waits for the LDS reads too) nobody reads the buffer any more
            if (t + 1 < s1) issue_stage(smem, t + 1);
            multiply(a1, b1);
        }
    } else {
        for (int t = s0; t < s1; ++t) {
            const int stage = (t - s0) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my DMAs of stage t have landed
            __syncthreads();                                            // ... everyone's have; stage^1 is no longer being read
            if (t + 1 < s1) issue_stage(smem + (stage ^ 1) * (OPA + OPB), t + 1);
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) {
                bf16x8 a[TI], b[4];
                load_frags(smem + stage * (OPA + OPB), ks, a, b);
                multiply(a, b);
            }
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
            float* cp = p.C + (size_t)blockIdx.y * p.c_slice + (long)row * p.ldc + col;
            float v[4] = {p.alpha * acc[i][j][0], p.alpha * acc[i][j][1], p.alpha * acc[i][j][2], p.alpha * acc[i][j][3]};
            const int nv = (p.N - col < 4) ? p.N - col : 4;
            const bool full = vec && nv == 4;
            if (p.beta != 0.f) {
                if (full) { const float4 c = *reinterpret_cast<const float4*>(cp); v[0] += p.beta * c.x; v[1] += p.beta * c.y; v[2] += p.beta * c.z; v[3] += p.beta * c.w; }
                else for (int r = 0; r < nv; ++r) v[r] += p.beta * cp[r];
            }
            const float r1 = p.r1row ? p.r1row[row] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < nv && p.bias) v[r] += p.bias[col + r];
                if (r < nv && p.r1row) v[r] = fmaf(r1, p.r1col[col + r], v[r]);
                if (p.act == FT_ACT_TANH) v[r] = tanhf_(v[r]);
                else if (p.act == FT_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
                else if (p.act == FT_ACT_SIGMOID) v[r] = sigmoidf_(v[r]);
            }
            if (p.c16) {
                unsigned short* hp = reinterpret_cast<unsigned short*>(p.C) + (long)row * p.ldc + col;
                if (full) *reinterpret_cast<uint2*>(hp) = make_uint2(pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3]));
                else for (int r = 0; r < nv; ++r) hp[r] = f2op16(v[r]);
            } else if (full) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int r = 0; r < nv; ++r) cp[r] = v[r];
        }
    }
}

// (Rounds 3-6 also carried a 256 x 256 x 64 tile for the k-contiguous x k-contiguous GEMMs with many tiles and a long reduction: 8 waves,
// 128 KiB of LDS, one workgroup per CU -- first as two wave groups half a phase apart with eight barriers per K-tile, then as a
// persistent kernel with one barrier per K-tile and the C stores of a tile issued behind the next tile's first unit.  Both reached
// 718-760 TFLOP/s at M 19 200 x N 4096 x K 1664 on N(0,1) data; the timing-only cuts (profiles/r06_gemm_bigk_cuts.log,
// r06_gemm_p256_cuts.log, r06_gemm_nostore.log, r06_gemm_pmc.log) say why: the 64 MFMAs of a K-tile are 0.85 us per SIMD pair and its
// 64 KB arrive in 1.2 us when nothing else runs, but with ONE workgroup per CU the L2 -> LDS stream drains at every barrier (waves in
// s_waitcnt / s_barrier 45 % of their cycles, TCP stalled on pending L2 returns 37 %) and the 256 KB of C per tile leave through the
// same in-order vmcnt the DMAs are waited on.  The single-buffer 128 x 128 x 64 form above -- four workgroups per CU, whole-line
// pieces -- is faster on the same shapes (759 / 798 with 16-bit C) and everywhere else, so the tile was removed.)

template <bool AKM, bool BKM>
void launch_s(const BfP& p, dim3 grid, bool big, bool wide, hipStream_t st) {
    const bool atomics = p.splits > 1 && p.c_slice == 0;       // (deterministic split-K runs the store epilogue, one C slice per k-slice)
    if (big) {
        if (atomics) hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, true, 256>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, false, 256>), grid, dim3(256), 0, st, p);
    } else {
        if (atomics) hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, true, 128>), grid, dim3(256), 0, st, p);
        else if constexpr (!(AKM && BKM)) {
            if (wide) hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, false, 128, 64, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, false, 128>), grid, dim3(256), 0, st, p);
        } else hipLaunchKernelGGL((gemm_bf16_k<AKM, BKM, false, 128>), grid, dim3(256), 0, st, p);
    }
}

// C[m][n] = sum_s work[s][m][n] (s ascending: a fixed association) + bias[n]; work = [S][M][N] fp32, N % 4 == 0
__global__ __launch_bounds__(256) void splitk_reduce_k(const float* __restrict__ work, int S, long MN, int N, float* __restrict__ C, long ldc,
                                                       const float* __restrict__ bias) {
    const long n4 = MN >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i << 2;
        const long m = e / N;
        const int n = (int)(e - m * N);
        float4 a = *reinterpret_cast<const float4*>(work + e);
        for (int sl = 1; sl < S; ++sl) {
            const float4 b = *reinterpret_cast<const float4*>(work + (size_t)sl * MN + e);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) { a.x += bias[n]; a.y += bias[n + 1]; a.z += bias[n + 2]; a.w += bias[n + 3]; }
        *reinterpret_cast<float4*>(C + m * ldc + n) = a;
    }
}

// tile height (256 x 128 workgroup tiles, 2 per CU, when they still fill the chip, possibly with split-K; else 128 x 128, 4 per CU)
// and the number of k-slices of a GEMM that may split its reduction
long plan_slices(int M, int N, int K, bool can_split, int compact, bool* big_out) {
    static const int force_tile = [] { const char* e = getenv("FT_GEMM_BF16_TILE"); return e ? atoi(e) : 0; }();
    const long tiles_big = (long)cdiv(M, 256) * cdiv(N, TB);
    // measured (scripts/exp/gemm_bench.py): the tall tile pays for the long-K weight-gradient shapes (+9 %), is neutral to
    // slightly slower for the forward / input-gradient shapes -- those keep the 128 x 128 tile at 4 workgroups per CU
    bool big = M >= 512 && can_split && tiles_big * (K / 512) >= 384;
    if (force_tile == 128) big = false;
    if (force_tile == 256) big = M >= 256;
    const int RTA = big ? 256 : TB;
    const long slots = big ? 512 : 1024;
    const long tiles = (long)cdiv(N, TB) * cdiv(M, RTA);
    long s = 1;
    if (can_split && tiles < slots / 2) {
        s = slots / tiles;                     // fill all workgroup slots of the 256 CUs
        const long smax = K / 512;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    if (compact == 1) s = 1;                   // scattered output rows: no split-K (the zero fill would have to be scattered too)
    // (compact reduction: K is the capacity; the kernel divides the rows the batch HAS evenly over the s slices.  Until round 6 the
    // slices divided the capacity and s was over-split by 4/3: dW_ih0 [4096 x 1664] 319 -> 273 us, the step's 18 tall split-K GEMMs
    // 2.72 -> 2.51 ms, profiles/r06_gemm_split_sweep.log)
    *big_out = big;
    return s;
}

// images -> C.  a_km / b_km: the operand image is k-major ([k][row]) instead of k-contiguous ([row][k]).
// Images are padded to multiples of 256 in both dimensions (ft_bf16_image), so either tile height may run off the logical M.
int run_images(const unsigned short* A, long lda, int a_km, const unsigned short* B, long ldb, int b_km, float* C, long ldc,
               const float* bias, int M, int N, int K, float alpha, float beta, int act, int flags, hipStream_t st,
               const int* rowmap = nullptr, const int* rows_dev = nullptr, int compact = 0, int k_shift = 0,
               const float* r1row = nullptr, const float* r1col = nullptr, float* split_work = nullptr, size_t split_work_bytes = 0) {
    BfP p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.nk = cdiv(K, 32); p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta; p.act = act;
    p.rowmap = rowmap; p.rows_dev = rows_dev; p.compact = compact; p.k_shift = k_shift;
    p.r1row = r1row; p.r1col = r1col;
    p.c_slice = 0;
    p.c16 = (flags & FT_GEMM_C16) ? 1 : 0;
    if (p.c16 && (beta != 0.f || (flags & (FT_GEMM_SPLITK | FT_GEMM_SPLITK_DET))))
        return ft_fail(FT_EINVAL, "ft_gemm_img: FT_GEMM_C16 takes beta == 0 and no split-K");
    // deterministic split-K: the slices' partial products side by side in a workspace + a fixed-order reduction (for FORWARD GEMMs
    // with few output tiles and a long K: a forward pass must be a function of its inputs, which the atomics' order is not)
    const bool det = (flags & FT_GEMM_SPLITK_DET) && split_work && act == FT_ACT_NONE && beta == 0.f && compact == 0 && !r1row && K >= 2048 &&
                     N % 4 == 0 && ldc % 4 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0 && reinterpret_cast<uintptr_t>(split_work) % 16 == 0;
    const bool can_split = det || ((flags & FT_GEMM_SPLITK) && act == FT_ACT_NONE && (beta == 0.f || beta == 1.f) && K >= 2048 && !r1row);
    bool big;
    long s = plan_slices(M, N, K, can_split, compact, &big);
    const int RTA = big ? 256 : TB;
    p.gx = cdiv(N, TB); p.gy = cdiv(M, RTA);
    p.vec_c = (reinterpret_cast<uintptr_t>(C) % 16 == 0 && ldc % 4 == 0) ? 1 : 0;      // (16-bit C: 8-byte pieces of rows that start 8-byte aligned)
    if (det) {                                   // as many slices as the workspace holds
        const long fit = (long)(split_work_bytes / ((size_t)M * N * sizeof(float)));
        if (s > fit) s = fit;
        if (s < 1) s = 1;
    }
    // 64-wide k stages in ONE LDS buffer for the store kernels with a k-contiguous operand (Operand64: whole-line DMA pieces; 32 KiB of
    // LDS, 4 workgroups per CU) -- where K is a whole number of them (else the last stage would read 32 columns past K, which only
    // images padded by ft_bf16_image hold as zeros; a view into a wider image does not).  Measured on N(0,1) data, M 19 200
    // (scripts/exp/gemm_img_bench.py / gemm_step_bench.py, profiles/r06_gemm_wide_stages.log; 32-wide double-buffered stages -> 64-wide
    // double-buffered at 2 workgroups per CU -> 64-wide single-buffered at 4): x[R,1664] W[4096,1664]^T 602 -> 648 -> 759 TFLOP/s
    // (the 256^2 kernels: 748), K 1024: 635 -> 599 -> 789, d[R,4096] W[4096,1664] (dX) 730 -> 884 -> 967, N = 1024 dense layers
    // 580-600 -> 690-700.  FT_GEMM_BF16_WIDE=0 (A/B hook, read per call): 32-wide stages everywhere
    const char* wide_env = getenv("FT_GEMM_BF16_WIDE");
    const bool wide = (!wide_env || atoi(wide_env) != 0) && (p.nk & 1) == 0 && !(a_km && b_km) && !big;
    p.ksteps = cdiv(p.nk, s);
    if (wide && (p.ksteps & 1)) ++p.ksteps;            // (k-slices of whole 64-wide stages)
    p.splits = cdiv(p.nk, p.ksteps);
    const bool det_on = det && p.splits > 1;
    if (det_on) {
        p.C = split_work; p.ldc = N; p.c_slice = (long)M * N; p.bias = nullptr;
        p.vec_c = 1;
    } else if (p.splits > 1 && beta == 0.f) {
        FT_CHECK_HIP(hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, st));
    }
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
    if (a_km) { if (b_km) launch_s<true, true>(p, grid, big, wide, st); else launch_s<true, false>(p, grid, big, wide, st); }
    else      { if (b_km) launch_s<false, true>(p, grid, big, wide, st); else launch_s<false, false>(p, grid, big, wide, st); }
    if (det_on) {
        const long n4 = ((long)M * N) >> 2;
        const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_reduce_k, dim3(blocks), dim3(256), 0, st, split_work, p.splits, (long)M * N, N, C, ldc, bias);
    }
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


// ---------------------------------------------------------------------------------------------------------------
// N = 1 projections over a compact image (the gate layer, flowtron.py:760-761, reads the SAME [h_att ; ctx] rows as the decoder
// LSTM's input projection): a GEMV over the image the GEMM has just used instead of two fp32 GEMMs with one output column, and
// its weight gradient as a GEMV^T.  Operand rounding as in every 16-bit GEMM of the library (x from the image, w rounded here),
// fp32 accumulation.
// ---------------------------------------------------------------------------------------------------------------
// one wave per compact row; the separator row of utterance b (its first padded frame, t == lens[b]) also writes its value to
// every later frame of b (all padded frames of an utterance hold the same inputs: ft_pad_rows_fill's copy_separator)
__global__ __launch_bounds__(256) void img_gemv_rows_k(const unsigned short* __restrict__ img, long ld, int K, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                       const int* __restrict__ rowmap, const int* __restrict__ rows_dev,
                                                       const int* __restrict__ lens, int T, int B) {
    extern __shared__ float s_w[];                       // ceil8(K) rounded weights
    const int K8 = (K + 7) & ~7;
    for (int k = threadIdx.x; k < K8; k += 256) s_w[k] = k < K ? op16_to_f(f2op16(w[k])) : 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nrows = rows_dev[0];
    const float b0 = bias ? bias[0] : 0.f;
    for (int c = blockIdx.x * 4 + wave; c < nrows; c += gridDim.x * 4) {
        const int dest = rowmap[c];
        if (dest < 0) continue;
        const unsigned short* row = img + (size_t)c * ld;
        float acc = 0.f;
        for (int k = lane * 8; k < K8; k += 512) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + k);
            const float4 wa = *reinterpret_cast<const float4*>(s_w + k), wb = *reinterpret_cast<const float4*>(s_w + k + 4);
            acc = fmaf(op16_to_f(v.x & 0xffffu), wa.x, acc); acc = fmaf(op16_to_f(v.x >> 16), wa.y, acc);
            acc = fmaf(op16_to_f(v.y & 0xffffu), wa.z, acc); acc = fmaf(op16_to_f(v.y >> 16), wa.w, acc);
            acc = fmaf(op16_to_f(v.z & 0xffffu), wb.x, acc); acc = fmaf(op16_to_f(v.z >> 16), wb.y, acc);
            acc = fmaf(op16_to_f(v.w & 0xffffu), wb.z, acc); acc = fmaf(op16_to_f(v.w >> 16), wb.w, acc);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        acc += b0;
        const int t = dest / B, b = dest - t * B;
        int len = lens[b];
        len = len < 0 ? 0 : len;
        if (lane == 0) y[(size_t)dest * ldy] = acc;
        if (t == len)                                                       // separator: stands for every padded frame of b
            for (int tt = t + 1 + lane; tt < T; tt += 64) y[((size_t)tt * B + b) * ldy] = acc;
    }
}

// dw[k] += sum_c dy[rowmap[c]] * img[c][k],  db += sum_c dy[rowmap[c]]  (dw, db zeroed by the caller); a block owns RPB compact rows,
// a thread 8 columns
template <int RPB>
__global__ __launch_bounds__(256) void img_gemv_rows_bwd_k(const unsigned short* __restrict__ img, long ld, int K, const float* __restrict__ dy,
                                                           long lddy, float* __restrict__ dw, float* __restrict__ db,
                                                           const int* __restrict__ rowmap, const int* __restrict__ rows_dev) {
    __shared__ float s_d[RPB];
    const int nrows = rows_dev[0];
    const int c0 = blockIdx.x * RPB;
    if (c0 >= nrows) return;
    const int nr = nrows - c0 < RPB ? nrows - c0 : RPB;
    for (int i = threadIdx.x; i < RPB; i += 256) {
        float d = 0.f;
        if (i < nr) { const int dest = rowmap[c0 + i]; if (dest >= 0) d = dy[(size_t)dest * lddy]; }
        s_d[i] = d;
    }
    __syncthreads();
    const int K8 = (K + 7) & ~7;
    for (int k = threadIdx.x * 8; k < K8; k += 2048) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned short* col = img + (size_t)c0 * ld + k;
        // eight rows in flight (rows at or beyond nr re-read the last valid row with a zero factor: the image is only defined up to
        // the zero rows that follow *rows_dev)
        for (int i0 = 0; i0 < nr; i0 += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < nr ? i0 + u : nr - 1;
                v[u] = *reinterpret_cast<const uint4*>(col + (size_t)i * ld);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = i0 + u < nr ? s_d[i0 + u] : 0.f;
                acc[0] = fmaf(d, op16_to_f(v[u].x & 0xffffu), acc[0]); acc[1] = fmaf(d, op16_to_f(v[u].x >> 16), acc[1]);
                acc[2] = fmaf(d, op16_to_f(v[u].y & 0xffffu), acc[2]); acc[3] = fmaf(d, op16_to_f(v[u].y >> 16), acc[3]);
                acc[4] = fmaf(d, op16_to_f(v[u].z & 0xffffu), acc[4]); acc[5] = fmaf(d, op16_to_f(v[u].z >> 16), acc[5]);
                acc[6] = fmaf(d, op16_to_f(v[u].w & 0xffffu), acc[6]); acc[7] = fmaf(d, op16_to_f(v[u].w >> 16), acc[7]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) if (k + e < K) atomicAdd(dw + k + e, acc[e]);
    }
    if (db && threadIdx.x < 64) {
        float sacc = 0.f;
        for (int i = threadIdx.x; i < nr; i += 64) sacc += s_d[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sacc += __shfl_xor(sacc, o);
        if (threadIdx.x == 0) atomicAdd(db, sacc);
    }
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

// dst: ft_bf16_image_bytes(rows, 3 * cols) bytes; row stride ceil256(3 * cols) elements
// n (<= 32) weight images in one launch: kind 0 = ft_bf16_image(src, ld, rows, cols, dst), 1 = ft_bf16_image_split3(src, ld, rows, cols,
// dst, weight = 1); descs is a HOST array
extern "C" int FT_OPNAME(ft_bf16_image_table)(const ft_img_desc* descs, int n, void* stream) {
    FT_CHECK_ARG(descs && n >= 1 && n <= IMG_TABLE_MAX);
    ImgTable t{};
    t.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ft_img_desc& d = descs[i];
        FT_CHECK_ARG(d.src && d.dst && d.rows >= 1 && d.cols >= 1 && d.ld >= d.cols && d.rows < (1ll << 31) - 256 && 3 * d.cols < (1ll << 31) - 256);
        FT_CHECK_ARG(reinterpret_cast<uintptr_t>(d.dst) % 256 == 0 && (d.kind == 0 || (d.kind == 1 && d.cols % 8 == 0)));
        t.src[i] = d.src; t.dst[i] = reinterpret_cast<unsigned short*>(d.dst); t.sr[i] = (long)d.ld;
        t.R[i] = (int)d.rows; t.Cc[i] = (int)d.cols; t.kind[i] = d.kind;
        t.Rp[i] = (int)up((size_t)d.rows + 32, 256);
        t.Cp[i] = (int)up((size_t)(d.kind == 1 ? 3 * d.cols : d.cols), 256);
        t.vec[i] = (reinterpret_cast<uintptr_t>(d.src) % 16 == 0 && d.ld % 4 == 0) ? 1 : 0;
        const size_t chunks = (size_t)t.Rp[i] * (t.Cp[i] >> 3);
        const int nb = (int)((chunks + 255) / 256 < 2048 ? (chunks + 255) / 256 : 2048);
        t.blk0[i] = blocks;
        blocks += nb;
    }
    t.blk0[n] = blocks;
    hipLaunchKernelGGL(img_table_k, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_bf16_image_split3)(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, int weight, void* stream) {
    FT_CHECK_ARG(src && dst && rows >= 1 && cols >= 8 && cols % 8 == 0 && ld >= cols && rows < (1ll << 31) - 256 && 3 * cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    const int Rp = (int)up((size_t)rows + 32, 256), Cp = (int)up((size_t)3 * cols, 256);
    const size_t chunks = (size_t)Rp * (Cp >> 3);
    const int blocks = (int)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
    hipLaunchKernelGGL(img_split3_k, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, (long)ld, (int)rows, (int)cols,
                       reinterpret_cast<unsigned short*>(dst), Rp, Cp, weight ? 1 : 0);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

extern "C" int FT_OPNAME(ft_bf16_image_split3_im2col)(const float* x, const int32_t* lens, int L, int B, int C, int KW, void* dst, void* stream) {
    FT_CHECK_ARG(x && lens && dst && L >= 1 && B >= 1 && C >= 1 && KW >= 1 && (KW & 1) && ((int64_t)C * KW) % 8 == 0);
    FT_CHECK_ARG((int64_t)L * B < (1ll << 31) - 256 && 3ll * C * KW < (1ll << 31) - 256 && reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    const int Rp = (int)up((size_t)L * B + 32, 256), Cp = (int)up((size_t)3 * C * KW, 256);
    const size_t chunks = (size_t)Rp * (Cp >> 3);
    const int blocks = (int)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
    hipLaunchKernelGGL(img_split3_im2col_k, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, lens, L, B, C, KW,
                       reinterpret_cast<unsigned short*>(dst), Rp, Cp);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

// (the _acc forms below ADD the column sums to a colsum the caller has zeroed -- a slice of the backward pass's zeroed slab -- instead
// of clearing it here: one memset dispatch less per bias gradient, 11 per training step)
static int image_colsum_impl(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, bool clear, void* stream) {
    FT_CHECK_ARG(src && dst && colsum && rows >= 1 && cols >= 1 && ld >= cols && rows < (1ll << 31) - 256 && cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Rp = (int)up((size_t)rows + 32, 256), Cp = (int)up((size_t)cols, 256);
    const int vec = (reinterpret_cast<uintptr_t>(src) % 16 == 0 && ld % 4 == 0) ? 1 : 0;
    if (clear) FT_CHECK_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)cols, st));
    hipLaunchKernelGGL(img_rows_sum_k, dim3(cdiv(Cp, 256), cdiv(Rp, IMG_SUM_SLAB)), dim3(256), 0, st, src, (long)ld, (int)rows, (int)cols,
                       reinterpret_cast<unsigned short*>(dst), Rp, Cp, vec, colsum, nullptr, nullptr);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int FT_OPNAME(ft_bf16_image_colsum)(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream) {
    return image_colsum_impl(src, ld, rows, cols, dst, colsum, true, stream);
}
extern "C" int FT_OPNAME(ft_bf16_image_colsum_acc)(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream) {
    return image_colsum_impl(src, ld, rows, cols, dst, colsum, false, stream);
}

#if FT_OPFMT == 0
// fp32 bytes of workspace a deterministic split-K ft_gemm_img of this shape can use (FT_GEMM_SPLITK_DET); 0: it would not split
extern "C" size_t ft_gemm_img_split_work_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 2048 || N % 4 != 0) return 0;
    bool big;
    const long s = plan_slices(M, N, K, true, 0, &big);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}
#endif

extern "C" int FT_OPNAME(ft_gemm_img)(const ft_gemm_img_args* a, void* stream) {
    FT_CHECK_ARG(a != nullptr);
    FT_CHECK_ARG(a->A && a->B && a->C && a->M >= 1 && a->N >= 1 && a->K >= 1);
    FT_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && reinterpret_cast<uintptr_t>(a->A) % 16 == 0 && reinterpret_cast<uintptr_t>(a->B) % 16 == 0);
    FT_CHECK_ARG(a->compact >= 0 && a->compact <= 2 && (a->compact == 0 || a->rows_dev) && (a->compact != 1 || a->rowmap));
    FT_CHECK_ARG(a->k_shift >= 0 && (a->compact == 2 || a->k_shift == 0));
    FT_CHECK_ARG((a->r1_row == nullptr) == (a->r1_col == nullptr));
    return run_images(reinterpret_cast<const unsigned short*>(a->A), a->lda, a->a_kmajor, reinterpret_cast<const unsigned short*>(a->B),
                      a->ldb, a->b_kmajor, a->C, a->ldc, a->bias, a->M, a->N, a->K, a->alpha, a->beta, a->act, a->flags,
                      reinterpret_cast<hipStream_t>(stream), a->rowmap, a->rows_dev, a->compact, a->k_shift, a->r1_row, a->r1_col,
                      reinterpret_cast<float*>(a->split_work), a->split_work_bytes);
}

// compact image: image row i = source row rowmap[i] (i < *rows_dev; -1 = zero row); buffer sized for cap_rows
// The output gradient of an ACTIVATED dense layer straight into its operand image: dst = image(dy * act'(pre)) over the compact
// rows of `rowmap`, with act' expressed through the saved output y (ft_act_bwd's formulas), colsum [cols] = the bias gradient.
// Replaces ft_act_bwd + ft_bf16_image_rows (one fp32 [rows, cols] tensor written and read back per dense layer).
static int image_rows_act_bwd_impl(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows,
                                   int64_t cols, void* dst, float* colsum, const int32_t* rowmap,
                                   const int32_t* rows_dev, bool clear, void* stream) {
    FT_CHECK_ARG(dy && y && dst && colsum && rowmap && rows_dev && cap_rows >= 1 && cols >= 1 && ld >= cols && ldy >= cols);
    FT_CHECK_ARG(act >= FT_ACT_NONE && act <= FT_ACT_SIGMOID && cap_rows < (1ll << 31) - 512 && cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Rp = (int)up((size_t)cap_rows + 32, 256), Cp = (int)up((size_t)cols, 256);
    const int vec = (reinterpret_cast<uintptr_t>(dy) % 16 == 0 && ld % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 && ldy % 4 == 0) ? 1 : 0;
    if (clear) FT_CHECK_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)cols, st));
    hipLaunchKernelGGL(img_rows_sum_k, dim3(cdiv(Cp, 256), cdiv(Rp, IMG_SUM_SLAB)), dim3(256), 0, st, dy, (long)ld, (int)cap_rows, (int)cols,
                       reinterpret_cast<unsigned short*>(dst), Rp, Cp, vec, colsum, rowmap, rows_dev, y, (long)ldy, act);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int FT_OPNAME(ft_bf16_image_rows_act_bwd)(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows,
                                                     int64_t cols, void* dst, float* colsum, const int32_t* rowmap,
                                                     const int32_t* rows_dev, void* stream) {
    return image_rows_act_bwd_impl(dy, ld, y, ldy, act, cap_rows, cols, dst, colsum, rowmap, rows_dev, true, stream);
}
extern "C" int FT_OPNAME(ft_bf16_image_rows_act_bwd_acc)(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows,
                                                         int64_t cols, void* dst, float* colsum, const int32_t* rowmap,
                                                         const int32_t* rows_dev, void* stream) {
    return image_rows_act_bwd_impl(dy, ld, y, ldy, act, cap_rows, cols, dst, colsum, rowmap, rows_dev, false, stream);
}

// One COLUMN BLOCK of a compact image (LinearFn over two inputs: [h_att ; ctx] -> one image, one K loop): the piece src [*, cols]
// goes to columns [col_off, col_off + cols) of the image dst (row stride dst_ld elements, sized by ft_bf16_image_bytes for the
// TOTAL width); columns up to col_off + fill_cols (>= cols, multiple of 8: pass the image's remaining width for the last piece)
// are zeroed.  Rows as ft_bf16_image_rows.
extern "C" int FT_OPNAME(ft_bf16_image_rows_into)(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, int64_t dst_ld,
                                                  int64_t col_off, int64_t fill_cols, const int32_t* rowmap, const int32_t* rows_dev,
                                                  void* stream) {
    FT_CHECK_ARG(src && dst && rowmap && rows_dev && cap_rows >= 1 && cols >= 1 && ld >= cols && cap_rows < (1ll << 31) - 512);
    FT_CHECK_ARG(col_off >= 0 && col_off % 8 == 0 && fill_cols >= cols && fill_cols % 8 == 0 && col_off + fill_cols <= dst_ld && dst_ld % 8 == 0);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    const int Rp = (int)up((size_t)cap_rows + 32, 256);
    make_image(src, ld, 1, (int)cap_rows, (int)cols, reinterpret_cast<unsigned short*>(dst) + col_off, Rp, (int)fill_cols,
               reinterpret_cast<hipStream_t>(stream), rowmap, rows_dev, (long)dst_ld);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

static int image_rows_impl(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                           const int32_t* rowmap, const int32_t* rows_dev, bool clear, void* stream) {
    FT_CHECK_ARG(src && dst && rowmap && rows_dev && cap_rows >= 1 && cols >= 1 && ld >= cols && cap_rows < (1ll << 31) - 512 && cols < (1ll << 31) - 256);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(dst) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Rp = (int)up((size_t)cap_rows + 32, 256), Cp = (int)up((size_t)cols, 256);
    if (!colsum) {
        make_image(src, ld, 1, (int)cap_rows, (int)cols, reinterpret_cast<unsigned short*>(dst), Rp, Cp, st, rowmap, rows_dev);
    } else {
        const int vec = (reinterpret_cast<uintptr_t>(src) % 16 == 0 && ld % 4 == 0) ? 1 : 0;
        if (clear) FT_CHECK_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)cols, st));
        hipLaunchKernelGGL(img_rows_sum_k, dim3(cdiv(Cp, 256), cdiv(Rp, IMG_SUM_SLAB)), dim3(256), 0, st, src, (long)ld, (int)cap_rows, (int)cols,
                           reinterpret_cast<unsigned short*>(dst), Rp, Cp, vec, colsum, rowmap, rows_dev);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int FT_OPNAME(ft_bf16_image_rows)(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                                             const int32_t* rowmap, const int32_t* rows_dev, void* stream) {
    return image_rows_impl(src, ld, cap_rows, cols, dst, colsum, rowmap, rows_dev, true, stream);
}
extern "C" int FT_OPNAME(ft_bf16_image_rows_acc)(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                                                 const int32_t* rowmap, const int32_t* rows_dev, void* stream) {
    return image_rows_impl(src, ld, cap_rows, cols, dst, colsum, rowmap, rows_dev, false, stream);
}

// y[rowmap[c]] = sum_k img[c][k] w[k] + bias[0] over the compact rows of an image (img_gemv_rows_k above); y [T*B] rows of stride ldy,
// padded frames receive their utterance's separator value.  The image must be zero beyond K up to the next multiple of 8 columns
// (ft_bf16_image_rows / _into pad with zeros).
extern "C" int FT_OPNAME(ft_img_gemv_rows)(const void* img, int64_t ld, int K, const float* w, const float* bias, float* y, int64_t ldy,
                                           const int32_t* rowmap, const int32_t* rows_dev, const int32_t* lens, int T, int B, void* stream) {
    FT_CHECK_ARG(img && w && y && rowmap && rows_dev && lens && K >= 1 && ld >= ((K + 7) & ~7) && ld % 8 == 0 && ldy >= 1 && T >= 1 && B >= 1);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(img) % 16 == 0 && K <= 16384);
    const int blocks = cdiv(T * B + B, 4) < 2048 ? cdiv(T * B + B, 4) : 2048;
    hipLaunchKernelGGL(img_gemv_rows_k, dim3(blocks), dim3(256), sizeof(float) * ((K + 7) & ~7), reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned short*>(img), (long)ld, K, w, bias, y, (long)ldy, rowmap, rows_dev, lens, T, B);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

// dw[k] += sum_c dy[rowmap[c]] img[c][k] (k < K), db[0] += sum_c dy[rowmap[c]] over the compact rows (separator rows included: a
// separator carries the gradient of its own frame, like the weight-gradient GEMMs over compact rows); dw / db accumulate -- the
// caller zeroes them; db may be NULL
extern "C" int FT_OPNAME(ft_img_gemv_rows_bwd)(const void* img, int64_t ld, int K, const float* dy, int64_t lddy, float* dw, float* db,
                                               const int32_t* rowmap, const int32_t* rows_dev, int64_t cap_rows, void* stream) {
    FT_CHECK_ARG(img && dy && dw && rowmap && rows_dev && K >= 1 && ld >= ((K + 7) & ~7) && ld % 8 == 0 && lddy >= 1 && cap_rows >= 1);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(img) % 16 == 0 && cap_rows < (1ll << 31) - 512);
    hipLaunchKernelGGL((img_gemv_rows_bwd_k<256>), dim3(cdiv((int)cap_rows, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned short*>(img), (long)ld, K, dy, (long)lddy, dw, db, rowmap, rows_dev);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
