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
    if (MODE == 1 || MODE == 3) for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(b[i]));
    __syncthreads();
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "a"(b[j]));
                else if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(b[j]), "v"(a));   // A operand from AGPRs (lstm_persist_fwd_ms_k)
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

// the M-split forward step's MFMA block (lstm_persist_fwd_ms_k): 64 A fragments in AGPRs, 32 B fragments read from LDS (16 in flight, one
// more behind every MFMA pair), 8 accumulators; MODE 0 = as in the kernel, 1 = all B reads up front (16 only), 2 = no LDS at all
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k3(float* out, long long* ticks, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char hs[4 * 2064];
    for (int i = threadIdx.x; i < 4 * 2064 / 4; i += 256) reinterpret_cast<unsigned*>(hs)[i] = 0x3f803f80u;
    u32x4 w[2][32];
    for (int tl = 0; tl < 2; ++tl) for (int c = 0; c < 32; ++c) w[tl][c] = (u32x4){(unsigned)tl, (unsigned)c, threadIdx.x, 0x3f803f80u};
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int c = 0; c < 32; ++c) asm volatile("" : "+a"(w[tl][c]));
    const int lane = threadIdx.x & 63, li = lane & 15, kg = lane >> 4;
    const unsigned char* hb_ = hs + (li & 3) * 2064 + kg * 16;
    float keep = 0.f;
    __syncthreads();
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        f32x4 acc[2][4];
        u32x4 bfr[32];
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 32; ++c) bfr[c] = (u32x4){(unsigned)c, 1u, 2u, (unsigned)it};
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) bfr[c] = *reinterpret_cast<const u32x4*>(hb_ + c * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int cb = MODE == 1 ? (c & 15) : c;
            if (c < 4) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[0][c]) : "a"(w[0][c]), "v"(bfr[cb]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[1][c]) : "a"(w[1][c]), "v"(bfr[cb]));
            } else {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0][c & 3]) : "a"(w[0][c]), "v"(bfr[cb]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[1][c & 3]) : "a"(w[1][c]), "v"(bfr[cb]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0 && c + 16 < 32) {
                bfr[c + 16] = *reinterpret_cast<const u32x4*>(hb_ + (c + 16) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        keep += acc[0][0][0] + acc[0][1][1] + acc[0][2][2] + acc[0][3][3] + acc[1][0][0] + acc[1][1][1] + acc[1][2][2] + acc[1][3][3];
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = keep;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* ticks;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
    long long h[256];
    for (int mode = 0; mode < 4; ++mode) for (int rep = 0; rep < 2; ++rep) {
        int iters = 2000;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("mode %d (%s): %.2f ns per MFMA per wave (64 MFMAs = %.3f us)\n", mode,
               mode == 0 ? "B in VGPR, acc VGPR" : mode == 1 ? "B in AGPR, acc VGPR" : mode == 2 ? "B in VGPR, acc AGPR" : "A in AGPR, B in VGPR, acc VGPR", avg * 10.0 / (iters * 64.0), avg * 10.0 / iters / 1000.0);
    }
    float* sink; hipMalloc(&sink, 1 << 22);
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        int iters = 2000;
        if (mode == 0) hipLaunchKernelGGL(k3<0>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else if (mode == 1) hipLaunchKernelGGL(k3<1>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else hipLaunchKernelGGL(k3<2>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("M-split MFMA block, %s: %.3f us per step (barrier + LDS reads + 64 MFMAs + 24 wait states + 8 adds)\n",
               mode == 0 ? "32 LDS fragment reads" : mode == 1 ? "16 LDS fragment reads up front" : "no LDS reads", avg * 10.0 / iters / 1000.0);
    }
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
