// ABI bookkeeping for libflowtron_hip.so
#include "common.h"

thread_local char g_ft_err[512] = {0};

extern "C" int ft_abi_version(void) { return FT_ABI_VERSION; }
extern "C" const char* ft_last_error(void) { return g_ft_err; }
