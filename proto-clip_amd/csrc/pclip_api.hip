// C-ABI plumbing shared by every entry point of libpclip: thread-local error text, launch checks,
// version / device queries.  No compute lives here.
#include "pclip_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void pclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pclip_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pclip_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return PCLIP_E_LAUNCH;
    }
    return PCLIP_OK;
}

extern "C" int pclip_abi_version(void) { return PCLIP_ABI_VERSION; }
extern "C" const char* pclip_last_error(void) { return g_err; }

extern "C" int pclip_device_cus(void) {
    static std::atomic<int> cache[64];                     // per device id: the tile dispatch asks on every launch
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64 && (cus = cache[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

