// Library-level entry points of include/d2p.h: version, error string, device info.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void d2p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int d2p_version(void) { return 1; }

extern "C" const char* d2p_last_error(void) { return g_err; }

extern "C" int d2p_device_info(int device, char* name, int name_len, int* cus, int* wave,
                               size_t* hbm_bytes) {
    hipDeviceProp_t prop;
    D2P_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (wave) *wave = prop.warpSize;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return D2P_OK;
}
