// hipemu -- a minimal single-threaded CPU emulation of the HIP device model, TEST INFRASTRUCTURE ONLY.
//
// The build container has no GPU and a round gets ~90 GPU-minutes, so the kernels of bbdm_amd/csrc/*.hip are also
// compiled for the host against this header (tools/hipemu/build.py) and executed on the CPU by a fiber scheduler that
// reproduces what the kernels rely on: 64-lane wavefronts, __syncthreads, dynamic / static LDS, __shfl_*,
// v_mfma_f32_32x32x2_f32 operand / accumulator lane maps, atomicAdd.  tests/test_emu_*.py run the C-ABI entry points
// of that host build on small shapes against PyTorch references before any GPU time is spent.
// Never shipped, never loaded by the product path (bbdm_amd/_lib.py loads libbbdm_hip.so only).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
// 8 "CUs": small enough that the test-sized by_batch GEMM launches walk several tiles per workgroup (gemm_bf3p.hip: persist)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 8; return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
// CU-partition streams (runtime.hip): the emulator hands out distinct dummy handles; every launch is synchronous anyway
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = malloc(1); return *s ? hipSuccess : 1; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3 };

namespace hipemu {
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern int g_lane;                    // lane of the running fiber within its wavefront
void* dyn_smem();                     // base of the dynamic LDS allocation of the running block
void block_sync();                    // __syncthreads
const void* const* wave_publish(const void* mine);   // publish a pointer, wave-sync, return the 64 published pointers
void wave_release();                  // second wave-sync: everybody has read, the published objects may die
void dma_issue(const char* gsrc_lane, char* lds_wave_base, int size);   // LDS-DMA, deferred (see the bottom of this header)
void dma_wait(int keep_in_flight);
typedef void (*body_fn)(void*);
void launch(dim3 grid, dim3 block, size_t shmem, body_fn fn, void* ctx);
template <class F>
inline void launch_l(dim3 grid, dim3 block, size_t shmem, F&& f) {
    launch(grid, block, shmem, [](void* c) { (*static_cast<typename std::remove_reference<F>::type*>(c))(); }, &f);
}
}  // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim
#define warpSize 64

// kernel<<<>>> is never used in this code base: every launch is hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...)
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                        \
    do {                                                                                               \
        (void)(stream);                                                                                \
        hipemu::launch_l(dim3(grid), dim3(block), (size_t)(lds), [&]() { (kern)(__VA_ARGS__); });      \
    } while (0)

static inline void __syncthreads() { hipemu::block_sync(); }

// ---- integer / float helpers HIP puts in the global namespace ------------------------------------------------------
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
// HIP's __expf: v_exp_f32 (2^x) of log2(e) * x, the constant as the device header has it (__clang_hip_math.h)
#define __expf(v) exp2f(0x1.715476p+0f * (v))
#define __logf(v) logf(v)
static inline float __frcp_rn(float v) { return 1.0f / v; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float v) { return 1.0f / sqrtf(v); }
static inline double rsqrt(double v) { return 1.0 / sqrt(v); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline long long wall_clock64() { return 0; }
static inline long long clock64() { return 0; }

// ---- atomics (one OS thread runs everything: plain read-modify-write) --------------------------------------------------
template <class T> static inline T hipemu_atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, float v) { return hipemu_atomic_add(p, v); }
static inline double atomicAdd(double* p, double v) { return hipemu_atomic_add(p, v); }
static inline int atomicAdd(int* p, int v) { return hipemu_atomic_add(p, v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return hipemu_atomic_add(p, v); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return hipemu_atomic_add(p, v); }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }

// ---- wavefront cross-lane operations ---------------------------------------------------------------------------------
template <class T>
static inline T hipemu_lane_read(T v, int src_lane) {
    const void* const* all = hipemu::wave_publish(&v);
    const void* q = (src_lane >= 0 && src_lane < 64) ? all[src_lane] : nullptr;
    T r = q ? *static_cast<const T*>(q) : v;          // inactive / out-of-range source: own value (as the hardware does)
    hipemu::wave_release();
    return r;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    const int l = hipemu::g_lane, s = l ^ mask;
    return hipemu_lane_read(v, (s / width == l / width) ? s : l);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int l = hipemu::g_lane, s = l + (int)d;
    return hipemu_lane_read(v, (s / width == l / width) ? s : l);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int l = hipemu::g_lane, s = l - (int)d;
    return hipemu_lane_read(v, (s >= 0 && s / width == l / width) ? s : l);
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    const int l = hipemu::g_lane;
    return hipemu_lane_read(v, (l / width) * width + (src % width));
}
static inline int hipemu_readfirstlane(int v) {
    const void* const* all = hipemu::wave_publish(&v);
    int r = v;
    for (int i = 0; i < 64; ++i) if (all[i]) { r = *static_cast<const int*>(all[i]); break; }
    hipemu::wave_release();
    return r;
}
static inline unsigned long long __ballot(int pred) {
    const void* const* all = hipemu::wave_publish(&pred);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (all[i] && *static_cast<const int*>(all[i])) m |= 1ull << i;
    hipemu::wave_release();
    return m;
}

// v_mfma_f32_32x32x2_f32: D[32x32] = A[32x2] * B[2x32] + C.  Lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32];
// register r of lane l holds D[(r % 4) + 8 * (r / 4) + 4 * (l / 32)][l % 32]  (CDNA ISA guide, 32x32 f32 layout).
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    struct AB { float a, b; } mine{a, b};
    const void* const* all = hipemu::wave_publish(&mine);
    const int l = hipemu::g_lane, j = l & 31, hi = l >> 5;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            const AB* pa = static_cast<const AB*>(all[i + 32 * k]);
            const AB* pb = static_cast<const AB*>(all[j + 32 * k]);
            acc = fmaf(pa ? pa->a : 0.f, pb ? pb->b : 0.f, acc);
        }
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}
// v_mfma_f32_16x16x4_f32: D[16x16] = A[16x4] * B[4x16] + C.  Lane l supplies A[l % 16][l / 16], B[l / 16][l % 16];
// register r of lane l holds D[4 * (l / 16) + r][l % 16].
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    struct AB { float a, b; } mine{a, b};
    const void* const* all = hipemu::wave_publish(&mine);
    const int l = hipemu::g_lane, j = l & 15, q = l >> 4;
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const AB* pa = static_cast<const AB*>(all[i + 16 * k]);
            const AB* pb = static_cast<const AB*>(all[j + 16 * k]);
            acc = fmaf(pa ? pa->a : 0.f, pb ? pb->b : 0.f, acc);
        }
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}
// v_mfma_f32_32x32x16_bf16: D[32x32] = A[32x16] * B[16x32] + C.  Lane l supplies A[l % 32][8 (l / 32) + e] and
// B[8 (l / 32) + e][l % 32], e = 0..7 (one 16-byte register quad each); C/D layout as the f32 32x32 form.
// Accumulation as the hardware does it (tools/microbench/mfma_round.hip -> profiles/r06_mfma_round.txt): the 16 products are added
// in TWO groups of 8 k (the two lane halves), each group summed exactly and added to the accumulator with one rounding.
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
template <typename V8>
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_16b(V8 a, V8 b, hipemu_f32x16 c) {
    struct AB { float a[8], b[8]; } mine;
    for (int e = 0; e < 8; ++e) { mine.a[e] = (float)a[e]; mine.b[e] = (float)b[e]; }
    const void* const* all = hipemu::wave_publish(&mine);
    const int l = hipemu::g_lane, j = l & 31, hi = l >> 5;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int h = 0; h < 2; ++h) {
            const AB* pa = static_cast<const AB*>(all[i + 32 * h]);
            const AB* pb = static_cast<const AB*>(all[j + 32 * h]);
            if (!pa || !pb) continue;
            double s = 0.0;
            for (int e = 0; e < 8; ++e) s += (double)pa->a[e] * (double)pb->b[e];
            acc = (float)((double)acc + s);
        }
        d[r] = acc;
    }
    hipemu::wave_release();
    return d;
}
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
    return hipemu_mfma_f32_32x32x16_16b(a, b, c);
}
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c, int, int, int) {
    return hipemu_mfma_f32_32x32x16_16b(a, b, c);
}
// ds_read_b64_tr_b16 (the LDS transpose read): every lane reads the 8-byte word (4 x 16 bit) at ITS address; inside each 16-lane
// group lane i receives element (i & 3) of the words of lanes 4 j + (i >> 2), j = 0..3 -- i.e. when the group's words form a
// row-major [4][16] matrix (lanes 0-3 row 0, lanes 4-7 row 1, ...), lane i gets column i (cdna_hip_programming.md, LDS section).
typedef short hipemu_v4s __attribute__((ext_vector_type(4)));
static inline hipemu_v4s hipemu_ds_read_tr16_b64(const void* p) {
    struct W { short e[4]; } mine;
    memcpy(&mine, p, 8);
    const void* const* all = hipemu::wave_publish(&mine);
    const int l = hipemu::g_lane, grp = l & ~15, i = l & 15;
    hipemu_v4s r;
    for (int j = 0; j < 4; ++j) {
        const W* w = static_cast<const W*>(all[grp + 4 * j + (i >> 2)]);
        r[j] = w ? w->e[i & 3] : (short)0x7fc0;
    }
    hipemu::wave_release();
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16_b64((const void*)(p))
// a wavefront executes in lockstep; the emulation's fibers do not: __builtin_amdgcn_wave_barrier() (a scheduling barrier on the GPU)
// is where the kernels say "every lane of the wave has executed what precedes" -- here a real rendezvous of the wave's fibers
static inline void hipemu_wave_barrier() { int dummy = 0; hipemu::wave_publish(&dummy); hipemu::wave_release(); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu_mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu_mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
#define __builtin_amdgcn_rcpf(v) (1.0f / (v))
#define __builtin_amdgcn_exp2f(v) exp2f(v)
#define BBDM_KEEP_IN_BRANCH(v) ((void)0)
#define __builtin_amdgcn_rsqf(v) (1.0f / sqrtf(v))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// csrc/embed.hip: four LDS reads + their wait in one asm statement
#define BBDM_LDS_READ4_1K(d0, d1, d2, d3, addr)                                                                         \
    do {                                                                                                                \
        const char* b__ = (const char*)hipemu::dyn_smem() + (addr);                                                     \
        d0 = *reinterpret_cast<const __typeof__(d0)*>(b__);                                                             \
        d1 = *reinterpret_cast<const __typeof__(d1)*>(b__ + 1024);                                                      \
        d2 = *reinterpret_cast<const __typeof__(d2)*>(b__ + 2048);                                                      \
        d3 = *reinterpret_cast<const __typeof__(d3)*>(b__ + 3072);                                                      \
    } while (0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
// csrc/lds_dma.h: the counted wait for LDS-DMA copies
#define BBDM_WAIT_VMCNT(N) hipemu::dma_wait(N)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0
// LDS-DMA (global_load_lds_dwordx4 & co): every lane fetches `size` bytes at its own global address; the wave's 64 pieces land
// at wave-uniform LDS base + lane * size.  Emulated as a DEFERRED copy: the destination is poisoned at issue and the bytes only
// arrive when the issuing thread executes a counted wait (s_waitcnt vmcnt(N) -> hipemu::dma_wait(N): all but the newest N of its
// copies complete) -- a fragment read that is not covered by a wait (+ barrier) sees NaNs, a copy issued over data another wave
// still reads destroys it, and a thread that ends with copies outstanding aborts.
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    hipemu::dma_issue((const char*)(g) + (off), (char*)(l) + (off), (size))
#define __builtin_amdgcn_s_barrier() hipemu::block_sync()
