// MFMA issue-rate probe (round 4): how long do 64 v_mfma_f32_16x16x32_bf16 of one wave per SIMD take, with the B operand in VGPRs vs AGPRs,
// at 256 workgroups x 4 waves (the persistent recurrences' geometry)?   hipcc --offload-arch=gfx950 -O2 -o scripts/exp/mfma_rate_probe scripts/exp/mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* ticks, int iters) {
    u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b[4] = {{1u, 2u, 3u, 4u}, {5u, 6u, 7u, 8u}, {9u, 1u, 2u, 3u}, {4u, 5u, 6u, 7u}};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (MODE == 1) for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(b[i]));
    __syncthreads();
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "a"(b[j]));
                else if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b[j]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b[j]));
            }
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
// the step's own shape: 64 fragments in AGPRs, 16 tiles x 4 chunks in groups of four, results tagged and stored (MODE 1 = with stores)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k2(float* out, long long* ticks, int iters, float* sink) {
    u32x4 w[16][4], a[4];
    for (int j = 0; j < 16; ++j) for (int g = 0; g < 4; ++g) { w[j][g] = (u32x4){(unsigned)j, (unsigned)g, threadIdx.x, 7u}; }
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+a"(w[j][g]));
    for (int g = 0; g < 4; ++g) a[g] = (u32x4){threadIdx.x, (unsigned)g, 2u, 3u};
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(sink, 0, 1 << 20, 0x00020000);
    const int voff = (threadIdx.x & 48) == 0 ? (int)((blockIdx.x * 16 + (threadIdx.x & 15)) * 16) : 0x7ffffff0;
    f32x4 keep = {0, 0, 0, 0};
    __syncthreads();
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        auto group = [&](int tq, f32x4 (&acc)[4]) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[jj]) : "v"(a[0]), "a"(w[tq * 4 + jj][0]));
#pragma unroll
            for (int g = 1; g < 4; ++g)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[jj]) : "v"(a[g]), "a"(w[tq * 4 + jj][g]));
        };
        auto publish = [&](int tq, const f32x4 (&acc)[4]) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                u32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (__float_as_uint(acc[jj][r]) & ~1u) | (unsigned)(it & 1);
                if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(v, wr, voff, (tq * 4 + jj) * 4096, 0);
                else keep[0] += __uint_as_float(v[0] ^ v[1] ^ v[2] ^ v[3]);
            }
        };
        f32x4 accA[4], accB[4];
        group(0, accA); group(1, accB); publish(0, accA); group(2, accA); publish(1, accB); group(3, accB); publish(2, accA);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        publish(3, accB);
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = keep[0];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* ticks;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
    long long h[256];
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        int iters = 2000;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("mode %d (%s): %.2f ns per MFMA per wave (64 MFMAs = %.3f us)\n", mode,
               mode == 0 ? "B in VGPR, acc VGPR" : mode == 1 ? "B in AGPR, acc VGPR" : "B in VGPR, acc AGPR", avg * 10.0 / (iters * 64.0), avg * 10.0 / iters / 1000.0);
    }
    float* sink; hipMalloc(&sink, 1 << 22);
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
        int iters = 2000;
        if (mode == 0) hipLaunchKernelGGL(k2<0>, dim3(256), dim3(256), 0, 0, out, ticks, iters, sink);
        else hipLaunchKernelGGL(k2<1>, dim3(256), dim3(256), 0, 0, out, ticks, iters, sink);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("step shape, %s: %.3f us per 64-MFMA step\n", mode == 0 ? "tags only" : "tags + 16 buffer stores", avg * 10.0 / iters / 1000.0);
    }
    return 0;
}
