// Shared helpers for the libbbdm_hip.so kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/bbdm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BBDM_WAVE 64

void bbdm_set_error(const char* fmt, ...);

// runtime.hip: the library's few integer options (bbdm_set_option / bbdm_get_option in the public header)
enum BbdmOption { BBDM_OPT_WGRAD1X1_BF3 = 0, BBDM_OPT_WINO_IDX64, BBDM_OPT_BF3P_KERNEL, BBDM_OPT_ATTN_BF3, BBDM_OPT_ATTN_PIPE, BBDM_OPT_BF3P_PAD_ROWS, BBDM_OPT_COUNT };
int bbdm_option(int id);

#define BBDM_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            bbdm_set_error(__VA_ARGS__);   \
            return BBDM_E_BADARG;          \
        }                                  \
    } while (0)

#define BBDM_CHECK_LAUNCH(what)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            bbdm_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));     \
            return BBDM_E_LAUNCH;                                                      \
        }                                                                              \
    } while (0)

// hipFuncSetAttribute applies to the CURRENT device: the "already raised the LDS limit" caches are kept per device
// (a process may drive several GPUs: the reference's `main.py --gpu_ids 1` runs on cuda:1 without set_device).
constexpr int BBDM_MAX_DEVICES = 64;
static inline int bbdm_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= BBDM_MAX_DEVICES) d = 0;
    return d;
}

// winograd_wgrad.hip: bbdm_gemm_tn_batched_f32 + the column sums of B per K split (internal; conv_wgrad.hip's 1x1 path)
int bbdm_gemm_tn_impl(const float* A, int lda, size_t a_stride, const float* B, int ldb, size_t b_stride, float* C, float* bias_part,
                      int batch, long long K, int M, int N, void* stream);

// Zero `bytes` (a multiple of 8, 8-byte aligned) on `st` with a KERNEL instead of hipMemsetAsync: the gradient plan is replayed as a
// hipGraph, and a captured memset node does not reproduce the eager memset on this ROCm (the GroupNorm-backward accumulators kept
// the previous replay's sums; found by bench.py's c4 parity check) -- kernel nodes do.
namespace {
__global__ void bbdm_zero8_kernel(unsigned long long* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0ull;
}
}  // namespace
static inline void bbdm_zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 8;
    if (!n) return;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bbdm_zero8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned long long*)p, n);
}

// Data that is written once and read once by the NEXT launch (the Winograd planes V and M: 0.5 - 7 GB per layer, many times the 256 MB
// Infinity Cache) moves with the `nt` policy: it does not displace the lines that are re-read (the halo columns of x, the weights).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(unsigned* p, unsigned v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float2 load_nt(const float2* p) {
    const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
    return make_float2(v.x, v.y);
}
// An empty asm that "modifies" v inside a rarely taken wave-uniform branch: without it the compiler if-converts the branch and runs its
// selects / multiplies in every iteration (tools/hipemu defines it away).
#ifndef BBDM_KEEP_IN_BRANCH
#define BBDM_KEEP_IN_BRANCH(v) asm volatile("" : "+v"(v))
#endif
#ifndef BBDM_NT_VSTORE
#define BBDM_NT_VSTORE 0
#endif
#ifndef BBDM_NT_MLOAD
#define BBDM_NT_MLOAD 0
#endif
#ifndef BBDM_NT_MSTORE
#define BBDM_NT_MSTORE 0
#endif


static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int ceil_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// conv_igemm.hip: `batch` independent GEMMs [H*W x CinPad] x [CinPad x Cout] (packed 1x1 weights) in one launch
int bbdm_conv1x1_batched(const float* x, int ldx, size_t xz, const float* packed_w, size_t wz, float* out, int ldo, size_t oz,
                         int batch, int H, int W, int CinPad, int Cout, hipStream_t st);

// gemm_bf3.hip (declared here for winograd.hip; also part of the public header)
extern "C" int bbdm_gemm_bf3_f32(const float* V, const void* packed_bf3, float* M, int batch, long long T, int CinPad, int Cout,
                                 void* stream);

// gemm_bf3p.hip (declared here for winograd.hip; also part of the public header)
extern "C" int bbdm_gemm_bf3p_f32(const void* a_planes, const void* b_planes, const float* bias, const float* residual, int ldr,
                                  float* M, int ldo, int batch, long long T, int CinPad, int Cout, void* stream);

// gemm_bf3p.hip: the fp16-pair GEMM whose A planes were scaled by (bound of the transform's input) x gain_a (winograd.hip)
int bbdm_gemm_h2p_gain_splitk(const void* a_planes, const void* b_planes, const float* bound_a, float gain_a, const float* bound_b,
                              float gain_b, float* M, int ldo, int batch, long long T, long long rows, int CinPad, int Cout, int splits,
                              void* stream);

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
// hardware v_exp_f32 + v_rcp_f32 (~2 ulp; __frcp_rn would expand to the full IEEE division sequence): for kernels where the exact-division form above would make an HBM-bound pass
// ALU-bound (the Winograd input transform evaluates each activation (m+2)^2/m^2 times)
__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// ... of a channel pair: the multiplies and the add as packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32), only the two
// v_exp_f32 and two v_rcp_f32 (quarter rate) per pair stay scalar; same values as silu_fast per element
__device__ __forceinline__ float2 silu_fast2(float2 v) {
    const f32x2 x = {v.x, v.y};
    const f32x2 t = x * -0x1.715476p+0f;                        // exp(-v) = 2^(-v log2 e): v_exp_f32 IS 2^x (__expf's constant)
    f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    const f32x2 o = x * r;
    return make_float2(o.x, o.y);
}

// ---- division of a non-negative index by a launch constant (Granlund-Montgomery): q = (mulhi(n, mul) + n) >> sh for 0 <= n < 2^31.
// A 32-bit division by a kernel ARGUMENT compiles to ~25 VALU instructions (float reciprocal + two correction steps); the input
// transform did two per lane to find its tile's (image, row, column).
struct FastDiv {
    unsigned d, mul, sh;
};
static inline FastDiv fastdiv_make(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned L = 0;
    while ((1ull << L) < d) ++L;
    f.mul = (unsigned)((((1ull << L) - d) << 32) / d + 1);
    f.sh = L;
    return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.sh; }
