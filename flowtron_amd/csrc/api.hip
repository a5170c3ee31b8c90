// ABI bookkeeping for libflowtron_hip.so
#include "common.h"

thread_local char g_ft_err[512] = {0};

extern "C" int ft_abi_version(void) { return FT_ABI_VERSION; }
extern "C" const char* ft_last_error(void) { return g_ft_err; }

// Test hook (tests/test_gpu_dist.py): `n_wg` workgroups that each claim a whole CU (all 160 KB of LDS) and spin for `ticks` of the
// 100 MHz wall clock -- what an in-flight collective kernel of another stream does to a whole-chip persistent launch: the persistent
// grid is not co-resident until they leave.  Bounded by construction (ticks is clamped to 5 s).
namespace {
__global__ __launch_bounds__(64) void hold_cus_k(long long ticks, int* sink) {
    extern __shared__ int big[];
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) { big[threadIdx.x] = ++n; __builtin_amdgcn_s_sleep(64); }
    if (sink && big[threadIdx.x] == -1) sink[0] = n;
}
}  // namespace
extern "C" int ft_debug_hold_cus(int n_wg, int64_t ticks, void* stream) {
    FT_CHECK_ARG(n_wg >= 1 && n_wg <= 256 && ticks >= 0);
    if (ticks > 500000000LL) ticks = 500000000LL;
    const int lds = 160 * 1024;                     // the whole LDS: nothing else fits on the CU
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hold_cus_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(hold_cus_k, dim3(n_wg), dim3(64), lds, reinterpret_cast<hipStream_t>(stream), (long long)ticks, (int*)nullptr);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
