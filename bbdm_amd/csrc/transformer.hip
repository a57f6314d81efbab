// transformer.hip -- the two token-wise ops of SpatialTransformer / BasicTransformerBlock that are not convolutions
// (model/BrownianBridge/base/modules/attention.py:196-219):
//   * nn.LayerNorm(dim) over the channel axis of each token (norm1 / norm2 / norm3, attention.py:204-206,215-217)
//   * GEGLU: x, gate = proj(x).chunk(2, -1); x * gelu(gate)          (attention.py:38-46; F.gelu = the exact erf form)
// Tokens are the pixels of the NHWC activation ('b c h w -> b (h w) c' is a no-op in this layout).  Both are HBM-bound
// streaming passes; the linear layers around them are 1x1 convolutions on the matrix core (csrc/conv_igemm.hip).
#include "common.h"

namespace {

// one wavefront per token: C / 64 elements per lane, two-pass (mean, then centred variance) in fp32 like at::layer_norm's
// CPU kernel (RowwiseMoments), wave reduction by DPP-free shuffles
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                        long long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= rows) return;                    // whole wavefronts leave together: no cross-lane op after this
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float a = v.x - mean, b = v.y - mean, d = v.z - mean, e = v.w - mean;
        q += (a * a + b * b) + (d * d + e * e);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    float* yr = y + (size_t)row * ldy;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 b = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v.x - mean) * rstd * g.x + b.x;
        o.y = (v.y - mean) * rstd * g.y + b.y;
        o.z = (v.z - mean) * rstd * g.z + b.z;
        o.w = (v.w - mean) * rstd * g.w + b.w;
        *reinterpret_cast<float4*>(yr + c) = o;
    }
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

__global__ void __launch_bounds__(256) geglu_kernel(const float* __restrict__ a, int lda, float* __restrict__ y, int ldy,
                                                    long long rows, int inner) {
    const int I4 = inner >> 2;
    const long long total = rows * I4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / I4;
        const int c = (int)(i - r * I4) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(a + (size_t)r * lda + c);
        const float4 gv = *reinterpret_cast<const float4*>(a + (size_t)r * lda + inner + c);
        float4 o;
        o.x = xv.x * gelu_erf(gv.x);
        o.y = xv.y * gelu_erf(gv.y);
        o.z = xv.z * gelu_erf(gv.z);
        o.w = xv.w * gelu_erf(gv.w);
        *reinterpret_cast<float4*>(y + (size_t)r * ldy + c) = o;
    }
}

}  // namespace

extern "C" int bbdm_layernorm_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy,
                                  long long rows, int C, float eps, void* stream) {
    BBDM_REQUIRE(x && gamma && beta && y && rows > 0 && C > 0, "layernorm: bad args");
    BBDM_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "layernorm: C=%d ldx=%d ldy=%d", C, ldx, ldy);
    BBDM_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm: 16-byte alignment");
    BBDM_REQUIRE((rows + 3) / 4 < (1ll << 31), "layernorm: too many rows");
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                       y, ldy, rows, C, eps);
    BBDM_CHECK_LAUNCH("layernorm");
    return BBDM_OK;
}

extern "C" int bbdm_geglu_f32(const float* a, int lda, float* y, int ldy, long long rows, int inner, void* stream) {
    BBDM_REQUIRE(a && y && rows > 0 && inner > 0, "geglu: bad args");
    BBDM_REQUIRE(inner % 4 == 0 && lda % 4 == 0 && ldy % 4 == 0 && lda >= 2 * inner && ldy >= inner, "geglu: inner=%d lda=%d ldy=%d",
                 inner, lda, ldy);
    BBDM_REQUIRE((((uintptr_t)a | (uintptr_t)y) & 15) == 0, "geglu: 16-byte alignment");
    long long blocks = (rows * (inner / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lda, y, ldy, rows, inner);
    BBDM_CHECK_LAUNCH("geglu");
    return BBDM_OK;
}
