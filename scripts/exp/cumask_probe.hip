// CU-mask probe (round 4): can a stream be confined to a subset of the 8 XCDs, and which bit of
// hipExtStreamCreateWithCUMask's mask is which (XCD, CU)?  Needed to decide whether recurrences of different LSTMs / GEMMs could
// run side by side on disjoint XCD sets.  Each workgroup records its XCC id and HW_ID, then spins ~50 us so
// that the whole grid is co-resident while it is observed.
//   build: hipcc --offload-arch=gfx950 -O2 -o scripts/exp/cumask_probe scripts/exp/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>
#include <set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { unsigned xcc, hwid; long long t0, t1; };

__global__ __launch_bounds__(256, 1) void probe_k(Rec* out, long long spin_ticks) {
    extern __shared__ float big[];                 // 100 KB of dynamic LDS: one workgroup per CU
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin_ticks) { big[0] = 1.f; }
        out[blockIdx.x] = Rec{xcc & 15u, hw, t0, wall_clock64()};
    }
}

static void report(const char* what, const std::vector<Rec>& r) {
    std::map<unsigned, std::set<unsigned>> cus;
    std::map<unsigned, int> cnt;
    long long tmin = r[0].t0, tmax = r[0].t1, tlate = r[0].t0;
    for (auto& x : r) {
        cus[x.xcc].insert((x.hwid >> 8) & 0xff);   // cu_id | sh_id | se_id bits
        cnt[x.xcc]++;
        if (x.t0 < tmin) tmin = x.t0;
        if (x.t1 > tmax) tmax = x.t1;
        if (x.t0 > tlate) tlate = x.t0;
    }
    printf("%-44s wgs %3zu  span %.1f us  last start +%.1f us | ", what, r.size(), (tmax - tmin) / 100.0, (tlate - tmin) / 100.0);
    for (auto& kv : cnt) printf("xcc%u: %d wg / %zu cu  ", kv.first, kv.second, cus[kv.first].size());
    printf("\n");
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("device CUs: %d\n", ncu);
    Rec* d;
    CK(hipMalloc(&d, sizeof(Rec) * 1024));
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto run = [&](hipStream_t st, int wgs, const char* what) {
        std::vector<Rec> h(wgs);
        CK(hipMemsetAsync(d, 0, sizeof(Rec) * 1024, st));
        hipLaunchKernelGGL(probe_k, dim3(wgs), dim3(256), lds, st, d, 5000LL);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), d, sizeof(Rec) * wgs, hipMemcpyDeviceToHost));
        report(what, h);
        return h;
    };
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    auto base = run(s0, 256, "no mask, 256 wgs");
    printf("  block -> xcc of the first 16 blocks:");
    for (int i = 0; i < 16; ++i) printf(" %u", base[i].xcc);
    printf("\n");
    run(s0, 128, "no mask, 128 wgs");
    run(s0, 32, "no mask, 32 wgs");

    struct Pat { const char* name; uint32_t m[8]; };
    std::vector<Pat> pats;
    { Pat p{"mask bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}}; pats.push_back(p); }
    { Pat p{"mask bits 32..63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0}}; pats.push_back(p); }
    { Pat p{"mask bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}}; pats.push_back(p); }
    { Pat p{"mask bits 128..255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}}; pats.push_back(p); }
    { Pat p{"mask bits i%8==0", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}}; pats.push_back(p); }
    { Pat p{"mask bits i%8==3", {0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u}}; pats.push_back(p); }
    { Pat p{"mask bits i%8<4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}}; pats.push_back(p); }
    { Pat p{"mask bits i%8>=4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}}; pats.push_back(p); }
    std::vector<hipStream_t> ms;
    for (auto& p : pats) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, p.m);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", p.name, hipGetErrorString(e)); ms.push_back(nullptr); continue; }
        ms.push_back(s);
        int nbits = 0;
        for (int i = 0; i < 8; ++i) nbits += __builtin_popcount(p.m[i]);
        char nm[128];
        snprintf(nm, sizeof nm, "%s, %d wgs", p.name, nbits);
        run(s, nbits, nm);
        snprintf(nm, sizeof nm, "%s, %d wgs (2x oversubscribed)", p.name, 2 * nbits);
        run(s, 2 * nbits, nm);
    }
    // concurrency: two half-chip streams at once (the two "i%8<4" / "i%8>=4" or "0..127" / "128..255" pairs): do both grids run
    // at the same time?  Records go to disjoint halves of the buffer.
    for (int pair = 0; pair < 2; ++pair) {
        hipStream_t a = ms[pair == 0 ? 2 : 6], b = ms[pair == 0 ? 3 : 7];
        if (!a || !b) continue;
        CK(hipDeviceSynchronize());
        CK(hipMemset(d, 0, sizeof(Rec) * 1024));
        hipLaunchKernelGGL(probe_k, dim3(128), dim3(256), lds, a, d, 20000LL);
        hipLaunchKernelGGL(probe_k, dim3(128), dim3(256), lds, b, d + 128, 20000LL);
        CK(hipDeviceSynchronize());
        std::vector<Rec> h(256);
        CK(hipMemcpy(h.data(), d, sizeof(Rec) * 256, hipMemcpyDeviceToHost));
        std::vector<Rec> ha(h.begin(), h.begin() + 128), hb(h.begin() + 128, h.end());
        report(pair == 0 ? "concurrent A (0..127)" : "concurrent A (i%8<4)", ha);
        report(pair == 0 ? "concurrent B (128..255)" : "concurrent B (i%8>=4)", hb);
        long long a0 = ha[0].t0, b0 = hb[0].t0;
        for (auto& x : ha) if (x.t0 < a0) a0 = x.t0;
        for (auto& x : hb) if (x.t0 < b0) b0 = x.t0;
        printf("  B started %.1f us after A (each spins 200 us: < 200 means they overlapped)\n", (b0 - a0) / 100.0);
    }
    return 0;
}
