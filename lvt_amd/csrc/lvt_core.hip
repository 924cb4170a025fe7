// Error reporting, version and device probe for liblvt_hip.so.
#include "lvt_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void lvt_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lvt_last_error(void) { return g_err; }
extern "C" int lvt_version(void) { return 610; }

extern "C" int lvt_device_info(char *name, int name_len, int *cus, int *clock_khz, long long *hbm_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        lvt_set_error("no HIP device visible");
        return LVT_ENODEVICE;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        lvt_set_error("hipGetDeviceProperties failed");
        return LVT_ENODEVICE;
    }
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (hbm_bytes) *hbm_bytes = (long long)prop.totalGlobalMem;
    return LVT_OK;
}
