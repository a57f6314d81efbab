// runtime.hip -- library-level entry points (version, per-thread error text, device properties).
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void bbdm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bbdm_version(void) { return 20; }
extern "C" const char* bbdm_last_error(void) { return g_err; }

namespace {
int device_cus() {
    static int cus_dev[BBDM_MAX_DEVICES] = {};
    int& cus = cus_dev[bbdm_device_slot()];
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}
}  // namespace

extern "C" int bbdm_device_cus(void) { return device_cus(); }
