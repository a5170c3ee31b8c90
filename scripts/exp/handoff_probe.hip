// Hand-off latency probe (MI355X): two workgroups ping-pong an 8-byte {epoch, value} granule through global memory.
// Reports ns per one-way hop for store flavour x load flavour x placement.  hipcc --offload-arch=gfx950 -O3 handoff_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int ST, int LD>
__global__ void pingpong(unsigned long long* buf, int partner_block, int iters, long* out, unsigned* xcc_out) {
    // block 0 = A, block partner_block = B, others idle
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
    if (threadIdx.x == 0 && me >= 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        xcc_out[me] = xcc;
    }
    if (me < 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    unsigned long long* mine = buf + me * 2048;        // my mailbox (partner writes it), 16 KB apart
    unsigned long long* theirs = buf + (1 - me) * 2048;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 512, 0x00020000);
    __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 16384, 0x00020000);
    long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (me == 0 || it > 0) {
            if (me == 0) {   // A sends first
                const unsigned long long g = ((unsigned long long)it << 32) | (unsigned)lane;
                if (lane < 16) {
                    if (ST == 0) theirs[lane] = g;
                    else if (ST == 1) __hip_atomic_store((gu64*)(theirs + lane), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store((gu64*)(theirs + lane), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // wait for epoch `it` in my mailbox (bounded: a flavour that never becomes visible must not hang the box)
            bool gave_up = false;
            for (unsigned spins = 0;; ++spins) {
                if (spins > (1u << 20)) { gave_up = true; break; }
                bool ok;
                if (LD == 0) {
                    const unsigned long long x = __hip_atomic_load((gu64*)(mine + (lane & 15)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)(x >> 32) == (unsigned)it;
                } else if (LD <= 2) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane & 7) * 16, 0, LD == 1 ? 2 : 16);
                    ok = v[1] == (unsigned)it && v[3] == (unsigned)it;
                } else {
                    // sweep-like pass: 16 x b128 per lane over 16 KB (only the first 128 B carry the granules), LD 3 = nt, 4 = sc1
                    u32x4 v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs16, (lane & 7) * 16, k * 1024, LD == 3 ? 2 : 16);
                    ok = v[0][1] == (unsigned)it && v[0][3] == (unsigned)it;
#pragma unroll
                    for (int k = 1; k < 16; ++k) ok &= (v[k][0] | 1u) != 0u;
                }
                asm volatile("" ::: "memory");          // the builtin loads are not atomics: keep them inside the spin loop
                if (__all(ok)) break;
            }
            if (gave_up) { if (lane == 0) out[1 + me] = it; break; }
            if (me == 1) {   // B replies
                const unsigned long long g = ((unsigned long long)it << 32) | (unsigned)lane;
                if (lane < 16) {
                    if (ST == 0) theirs[lane] = g;
                    else if (ST == 1) __hip_atomic_store((gu64*)(theirs + lane), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store((gu64*)(theirs + lane), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (lane == 0 && me == 0) out[0] = wall_clock64() - t0;
}

template <int ST, int LD>
void run(const char* name, int partner) {
    unsigned long long* buf; long* out; unsigned* xcc;
    hipMalloc(&buf, 32768); hipMalloc(&out, 64); hipMalloc(&xcc, 64);
    hipMemset(buf, 0, 32768); hipMemset(xcc, 0, 64); hipMemset(out, 0, 64);
    const int iters = 2000;
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(256), dim3(64), 0, 0, buf, partner, iters, out, xcc);
    hipError_t e = hipDeviceSynchronize();
    long res[3] = {0, 0, 0}; unsigned hx[2] = {0, 0};
    hipMemcpy(res, out, 24, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    if (res[1] || res[2]) printf("%-44s partner block %3d (xcc %u -> %u): NEVER VISIBLE (gave up at iteration %ld / %ld)\n", name, partner, hx[0], hx[1], res[1], res[2]);
    else printf("%-44s partner block %3d (xcc %u -> %u): %7.1f ns per hop   [%s]\n", name, partner, hx[0], hx[1], res[0] * 10.0 / (2.0 * iters),
                hipGetErrorString(e));
    fflush(stdout);
    hipFree(buf); hipFree(out); hipFree(xcc);
}

int main() {
    for (int partner : {8, 1}) {      // block 8: same XCD as block 0 (observed b % 8), block 1: another XCD
        run<2, 0>("sc1 (agent) store/ sc1 dwordx2 load", partner);
        run<2, 2>("sc1 (agent) store/ sc1 b128 load", partner);
        run<1, 0>("sc0 (wg) store   / sc1 dwordx2 load", partner);
        run<1, 2>("sc0 (wg) store   / sc1 b128 load", partner);
        run<1, 1>("sc0 (wg) store   / nt b128 load", partner);
        run<1, 3>("sc0 (wg) store   / 16 x nt b128 per poll", partner);
        run<1, 4>("sc0 (wg) store   / 16 x sc1 b128 per poll", partner);
        run<2, 4>("sc1 (agent) store/ 16 x sc1 b128 per poll", partner);
        run<0, 0>("plain store      / sc1 dwordx2 load", partner);
        run<0, 1>("plain store      / nt b128 load", partner);
    }
    return 0;
}
