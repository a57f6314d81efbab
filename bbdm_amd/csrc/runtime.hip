// runtime.hip -- library-level entry points (version, per-thread error text).
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void bbdm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bbdm_version(void) { return 12; }
extern "C" const char* bbdm_last_error(void) { return g_err; }
