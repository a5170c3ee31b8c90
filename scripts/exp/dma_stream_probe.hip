// L2 -> LDS streaming probe (round 6): how fast do the CUs pull operand images into LDS with global_load_lds_dwordx4, as a function of
//   * the piece shape: 16 rows x 64 B (the 32-wide k stages of gemm_bf16_k: half-line requests) against 8 rows x 128 B (64-wide k stages),
//   * the number of workgroups per CU (waves issuing), and the pipeline depth (pieces a wave keeps in flight across its wait).
// Every XCD's workgroups stream the same 2 MiB (L2-resident) region with a 3 328-byte row pitch (K = 1664 images), each from its own
// offset.  hipcc --offload-arch=gfx950 -O2 -o /tmp/dma_stream_probe scripts/exp/dma_stream_probe.hip && /tmp/dma_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// PIECES per wave and stage; ROWB = bytes per image row inside a piece (64 or 128); DEPTH = stages in flight (1 or 2)
template <int ROWB, int PIECES, int DEPTH>
__global__ __launch_bounds__(256) void stream_k(const unsigned char* img, long pitch, int rows_region, int iters, int lds_stage_bytes) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3;
    constexpr int LPR = ROWB / 16;                         // lanes per row
    constexpr int RPP = 64 / LPR;                          // rows per piece
    const unsigned char* base = img + (size_t)xcd * rows_region * pitch;
    int row = (wg * 4 + wave) * RPP * PIECES % rows_region;
    int col = 0;
    const int cols = (int)(pitch / ROWB);
    for (int it = 0; it < iters + DEPTH - 1; ++it) {
        if (it < iters) {
            unsigned char* st = smem + (it % DEPTH) * lds_stage_bytes + wave * PIECES * 1024;
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const int r = (row + p * RPP + lane / LPR) % rows_region;
                __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r * pitch + (size_t)col * ROWB + (lane % LPR) * 16),
                                                 (lds_void*)(st + p * 1024), 16, 0, 0);
            }
            col += 1;
            if (col == cols) { col = 0; row = (row + 4 * RPP * PIECES * 7) % rows_region; }
        }
        if (DEPTH == 1 || it + 1 >= iters) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <int ROWB, int PIECES, int DEPTH>
static void run(const unsigned char* img, long pitch, int rows_region, int wgs_per_cu, const char* name) {
    const int iters = 2000;
    const int stage = 4 * PIECES * 1024;
    const int lds = stage * DEPTH;
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_k<ROWB, PIECES, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // pad the dynamic LDS so that exactly wgs_per_cu workgroups fit a CU
    int pad = 160 * 1024 / wgs_per_cu;
    pad = pad / 1024 * 1024;
    if (pad < lds) pad = lds;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((stream_k<ROWB, PIECES, DEPTH>), dim3(grid), dim3(256), pad, 0, img, pitch, rows_region, 50, stage);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_k<ROWB, PIECES, DEPTH>), dim3(grid), dim3(256), pad, 0, img, pitch, rows_region, iters, stage);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * stage;
    printf("%-44s %d wg/cu  %7.3f ms  %6.2f TB/s  (%5.1f GB/s per CU)\n", name, wgs_per_cu, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const long pitch = 3328;
    const int rows_region = 640;                            // 640 rows x 3 328 B = 2.1 MB per XCD
    unsigned char* img;
    hipMalloc(&img, (size_t)8 * rows_region * pitch + 65536);
    hipMemset(img, 1, (size_t)8 * rows_region * pitch + 65536);
    for (int w = 1; w <= 4; ++w) {
        run<64, 4, 1>(img, pitch, rows_region, w, "16 rows x 64 B, 4 pieces / wave, depth 1");
        run<128, 4, 1>(img, pitch, rows_region, w, " 8 rows x 128 B, 4 pieces / wave, depth 1");
        run<64, 4, 2>(img, pitch, rows_region, w, "16 rows x 64 B, 4 pieces / wave, depth 2");
        run<128, 4, 2>(img, pitch, rows_region, w, " 8 rows x 128 B, 4 pieces / wave, depth 2");
        if (w <= 2) {
            run<64, 8, 1>(img, pitch, rows_region, w, "16 rows x 64 B, 8 pieces / wave, depth 1");
            run<128, 8, 1>(img, pitch, rows_region, w, " 8 rows x 128 B, 8 pieces / wave, depth 1");
            run<64, 8, 2>(img, pitch, rows_region, w, "16 rows x 64 B, 8 pieces / wave, depth 2");
            run<128, 8, 2>(img, pitch, rows_region, w, " 8 rows x 128 B, 8 pieces / wave, depth 2");
        }
    }
    return 0;
}
