// bf3_split.h -- the exact three-way bf16 split behind the fp32-accurate BF16-matrix-core products (gemm_bf3.hip: the Winograd
// tile GEMMs and wide 1x1 layers; attention.hip: Q K^T).  x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2) (round to nearest even; the residuals are exact in fp32), and a product keeps the six terms
// x1y1 + x1y2 + x2y1 + x1y3 + x3y1 + x2y2 (the dropped ones are below 2^-24 of |x||y|), accumulated in fp32 by the MFMA.
#pragma once
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_round(float x) {       // x rounded to bf16 (RNE), returned as fp32
    return (float)(__bf16)x;
}
// two fp32 -> two bf16 (RNE) packed in one dword: ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }

// a pair of fp32 -> the pair's three packed bf16 planes (11 VALU ops per pair)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = cvt_pk(x0, x1);
    const float r0 = x0 - lo_as_f32(p1), r1 = x1 - hi_as_f32(p1);          // exact
    p2 = cvt_pk(r0, r1);
    p3 = cvt_pk(r0 - lo_as_f32(p2), r1 - hi_as_f32(p2));
}

// split2 in two halves (the same operations: bit-equal), for loops that deal the split between their MFMAs (attention.hip)
__device__ __forceinline__ void split2_a(float x0, float x1, unsigned& p1, float& r0, float& r1) {
    p1 = cvt_pk(x0, x1);
    r0 = x0 - lo_as_f32(p1);
    r1 = x1 - hi_as_f32(p1);
}
__device__ __forceinline__ void split2_b(float r0, float r1, unsigned& p2, unsigned& p3) {
    p2 = cvt_pk(r0, r1);
    p3 = cvt_pk(r0 - lo_as_f32(p2), r1 - hi_as_f32(p2));
}

// a 4-element fp32 group -> 3 x (4 bf16 = 8 B)
__device__ __forceinline__ void split4(float4 v, uint2& p1, uint2& p2, uint2& p3) {
    split2(v.x, v.y, p1.x, p2.x, p3.x);
    split2(v.z, v.w, p1.y, p2.y, p3.y);
}

// which plane of A / B each of the six product terms takes, smallest terms first (they meet an accumulator not yet grown by
// the leading term): (x2,y2) (x1,y3) (x3,y1) (x1,y2) (x2,y1) (x1,y1)
constexpr int BF3_TA[6] = {1, 0, 2, 0, 1, 0}, BF3_TB[6] = {1, 2, 0, 1, 0, 0};

}  // namespace
