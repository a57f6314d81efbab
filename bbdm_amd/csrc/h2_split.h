// h2_split.h -- fp32 operands as TWO fp16 planes under a power-of-two scale (round 6): the arithmetic of the "h2" plane GEMMs.
//
// x * 2^e = h1 + h2 + r with h1 = fp16(x * 2^e), h2 = fp16(x * 2^e - h1) (round to nearest even; the residual is exact in fp32):
// 11 + 11 significand bits and the sign of h2, i.e. |r| <= 2^-23 |x * 2^e| as long as h2 is a normal fp16 number -- one bit short of
// fp32's own 2^-24.  A product keeps the three terms h1 k1 + h1 k2 + h2 k1 (the dropped h2 k2 is below 2^-22 of |x||y|), each an
// fp16 x fp16 product that is exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_f16 -- THREE matrix instructions per 16 k
// where the bf16x3 path (bf3_split.h) issues six.
//
// Why this is not a reduced-precision mode.  What limits an fp32 GEMM on the matrix core is not the 2^-24 of its operands but the
// chain of roundings of its fp32 accumulator: one per MFMA, each half an ulp of the running sum.  bf16x3 performs 6 K / 16 of them per
// output, the f32 MFMA (v_mfma_f32_32x32x2_f32) K / 2, this path 3 K / 16.  Measured against fp64 (tests/test_kernels_gpu.py::
// test_gemm_h2p_accuracy, tools/gemm_error_probe.py -> profiles/r06_gemm_error.txt) the h2 product is MORE accurate than either from
// K = 128 up, and through the ten-point Winograd transforms -- which amplify exactly that accumulated error -- it lowers the step's
// parity error.
//
// The scale.  fp16 has 5 exponent bits: normal numbers 2^-14 .. 65504.  h1 keeps 11 bits for |x 2^e| >= 2^-14 and h2 keeps its 11 for
// |x 2^e| >= 2^-3 (below that h2 turns subnormal and the ABSOLUTE error stays at 2^-25: harmless in a sum whose large terms carry
// 2^-23 relative).  So e must (a) never let |x 2^e| reach 65504 -- an overflow would be an inf in the result -- and (b) keep typical
// values above 2^-3.  It is derived from a BOUND of the tensor, `bound >= max |x|`, by one rule every producer and consumer evaluates
// for itself (h2_exp_of_bound): bound * 2^e lies in [2^14, 2^15).  A bound is provable or measured, never guessed:
//   * weights (U = G g G^T, 1x1 filters): the exact maximum, taken when the planes are packed (bf3p_absmax_kernel);
//   * activations behind GroupNorm: |gamma (1 + s)| sqrt(n_g - 1) + |beta (1 + s) + t| per (image, channel) -- a z-score cannot exceed
//     sqrt(n - 1) -- times the transform's gain (groupnorm.hip: the coefficient kernel takes the maximum; winograd.hip).
// With n_g up to 2^18 elements per group the bound sits ~2^12 above typical values, which leaves them 2^14 / 2^12 = 4 >> 2^-3.
#pragma once
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// e with bound * 2^e in [2^14, 2^15) (bound = 1.f x 2^E -> e = 14 - E).  A zero / subnormal bound (an all-zero tensor) and a
// non-finite one (poisoned input: the inf / nan reaches the result through the planes anyway) are clamped to a scale that is exact.
__host__ __device__ __forceinline__ int h2_exp_of_bound(float bound) {
    unsigned bits;
    memcpy(&bits, &bound, 4);
    int e = 14 - ((int)((bits >> 23) & 0xff) - 127);
    if (e > 60) e = 60;
    if (e < -60) e = -60;
    return e;
}
// 2^e as an fp32 number, -126 <= e <= 127
__host__ __device__ __forceinline__ float h2_pow2(int e) {
    const unsigned bits = (unsigned)(127 + e) << 23;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

// two fp32 -> two fp16 (RNE) packed in one dword: ONE v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned cvt_pk_h(float lo, float hi) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// a pair of (already scaled) fp32 -> the pair's two packed fp16 planes
__device__ __forceinline__ void h2_split2(float x0, float x1, unsigned& p1, unsigned& p2) {
    p1 = cvt_pk_h(x0, x1);
    const f16x2 h = __builtin_bit_cast(f16x2, p1);
    p2 = cvt_pk_h(x0 - (float)h.x, x1 - (float)h.y);          // (the differences are exact)
}
__device__ __forceinline__ void h2_split4(float4 v, uint2& p1, uint2& p2) {
    h2_split2(v.x, v.y, p1.x, p2.x);
    h2_split2(v.z, v.w, p1.y, p2.y);
}

}  // namespace
