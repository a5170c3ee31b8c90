// probe of ds_read_b64_tr_b16 on gfx950: LDS ushort[i] = i; lane l supplies byte address addr[l]; prints the 4 elements each lane receives
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned int a = (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned short*)lds + addr[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    int h[64]; int* d; unsigned short* o; unsigned short ho[256];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h[l] = l * 8;                               // lane-linear 8 B
            if (pat == 1) h[l] = (l & 15) * 64 + (l >> 4) * 8;        // 16 rows of 64 B, lane-group picks the 8-B column
            if (pat == 2) h[l] = ((l & 3) * 32 + (l >> 2 & 3) * 8) + (l >> 4) * 256;   // [4 k][16 m] blocks row stride 32 B
        }
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d, o);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d(elt %4d): %4d %4d %4d %4d\n", l, h[l], h[l] / 2, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
    }
    return 0;
}
