// Error reporting + version for the C ABI (include/gaot3d_hip.h).
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void gaot_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gaot_last_error(void) { return g_err; }
extern "C" int gaot_abi_version(void) { return GAOT_ABI_VERSION; }

long long g_gaot_launches = 0;
extern "C" int64_t gaot_launch_count(int reset) {
    const long long n = g_gaot_launches;
    if (reset) g_gaot_launches = 0;
    return (int64_t)n;
}
