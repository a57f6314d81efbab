// runtime.hip -- library-level entry points (version, per-thread error text, CU-partitioned streams).
#include <stdarg.h>
#include <map>
#include <mutex>
#include "common.h"

static thread_local char g_err[512] = "";

void bbdm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bbdm_version(void) { return 16; }
extern "C" const char* bbdm_last_error(void) { return g_err; }

// ---- CU-partitioned streams ---------------------------------------------------------------------------------------------------------
// A sampling step alternates MFMA-bound launches (Winograd tile GEMMs, 1x1 GEMMs, attention) with HBM-bound ones (Winograd transforms,
// GroupNorm passes): each kind leaves the other resource idle, and the matrix kernels are partly POWER-bound (profiles/
// r03_gemm_cu_partition_probe.txt: 25 % fewer CUs cost the tile GEMM 18 %).  bbdm_amd/unet.py therefore runs the two halves of a batch
// as two chains whose matrix launches and streaming launches go to two streams that own DISJOINT sets of CUs, so that one half's
// transforms stream through HBM while the other half's GEMMs keep the matrix cores busy.  A partition is a HIP stream created with a CU
// mask (hipExtStreamCreateWithCUMask).  KFD deals mask bit i to XCD i % 8 and walks the shader engines of that XCD with the bits it
// gets (mqd_symmetrically_map_cu_mask), so a bit range [8 a, 8 b) is b - a CUs on EVERY XCD -- each XCD keeps its share of both
// partitions, and the workgroup -> XCD round robin that the XCD-aware tile orders rely on is unchanged.
// Persistent kernels ask bbdm_stream_cus() how many CUs their stream owns and size their grids for it.
namespace {
std::mutex g_streams_mu;
std::map<void*, int> g_stream_cus;      // partition stream -> its CU count

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

extern "C" int bbdm_stream_create_partition(int cu_begin, int cu_end, void** stream) {
    const int cus = device_cus();
    BBDM_REQUIRE(stream && cu_begin >= 0 && cu_begin < cu_end && cu_end <= cus, "stream_create_partition: CU range [%d, %d) of %d CUs",
                 cu_begin, cu_end, cus);
    BBDM_REQUIRE(cu_begin % 8 == 0 && cu_end % 8 == 0, "stream_create_partition: [%d, %d) must be multiples of 8 (one bit per XCD in turn)",
                 cu_begin, cu_end);
    uint32_t mask[BBDM_MAX_CU_WORDS] = {};
    BBDM_REQUIRE(cus <= 32 * BBDM_MAX_CU_WORDS, "stream_create_partition: %d CUs", cus);
    for (int i = cu_begin; i < cu_end; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask);
    if (e != hipSuccess) {
        bbdm_set_error("stream_create_partition: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        return BBDM_E_LAUNCH;
    }
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        g_stream_cus[(void*)st] = cu_end - cu_begin;
    }
    *stream = (void*)st;
    return BBDM_OK;
}

extern "C" int bbdm_stream_destroy(void* stream) {
    BBDM_REQUIRE(stream, "stream_destroy: null stream");
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        const auto it = g_stream_cus.find(stream);
        BBDM_REQUIRE(it != g_stream_cus.end(), "stream_destroy: not a stream of bbdm_stream_create_partition");
        g_stream_cus.erase(it);
    }
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        bbdm_set_error("stream_destroy: %s", hipGetErrorString(e));
        return BBDM_E_LAUNCH;
    }
    return BBDM_OK;
}

// CUs the kernels enqueued on `stream` can run on: the partition's count, or the whole device for any other stream.
extern "C" int bbdm_stream_cus(void* stream) {
    if (stream) {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        const auto it = g_stream_cus.find(stream);
        if (it != g_stream_cus.end()) return it->second;
    }
    return device_cus();
}
