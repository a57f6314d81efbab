// runtime.hip -- library-level entry points (version, per-thread error text, device properties).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void bbdm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bbdm_version(void) { return 25; }

// ---- options: the few integer switches tests and tools/ flip (kernel A/B inside ONE process).  Not read from the environment, not
// latched: a launcher reads its option at every call.  Everything else the library decides from the shapes it is given.
namespace {
struct Option { const char* name; int value; };
Option g_options[BBDM_OPT_COUNT] = {
    {"wgrad1x1_bf3", 1},      // 1x1 weight gradients: 1 = the bf16x3 planes path from ~100 FLOP per split byte, 0 = always gemm_tn_f32,
                              // 2 = every shape the plane layout takes (tests)
    {"wino_idx64", 0},        // 1 = force the 64-bit row-address variant of the Winograd input transform (tests)
    {"bf3p_kernel", 6},       // tile shape of the pre-split GEMM: 6 = the library's choice, 4 / 5 / 7 = force 256x256 / 256x128 / 128x128
    {"attn_bf3", 1},          // attention forward: 1 = Q K^T and P V on the bf16x3 path, 2 = only Q K^T, 0 = both on the f32 MFMA (A/B)
    {"attn_pipe", 2},         // attention forward, full bf16x3 path: 1 = the loop with the splits dealt between the MFMAs, 0 = phased (A/B), 2 = 1 + K / V pre-split once per head for long sequences
    {"bf3p_pad_rows", 1},     // pre-split GEMM, ragged last row tile: 1 = its idle 32-row blocks read the producer's zero rows, 0 = the last real rows again (A/B)
};
int option_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < BBDM_OPT_COUNT; ++i)
        if (strcmp(name, g_options[i].name) == 0) return i;
    return -1;
}
}  // namespace
int bbdm_option(int id) { return g_options[id].value; }
extern "C" int bbdm_set_option(const char* name, int value) {
    const int i = option_index(name);
    BBDM_REQUIRE(i >= 0, "bbdm_set_option: unknown option '%s'", name ? name : "(null)");
    g_options[i].value = value;
    return BBDM_OK;
}
extern "C" int bbdm_get_option(const char* name, int* value) {
    const int i = option_index(name);
    BBDM_REQUIRE(i >= 0 && value, "bbdm_get_option: unknown option '%s' / null result", name ? name : "(null)");
    *value = g_options[i].value;
    return BBDM_OK;
}
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
