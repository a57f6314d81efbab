// transformer.hip -- the two token-wise ops of SpatialTransformer / BasicTransformerBlock that are not convolutions
// (model/BrownianBridge/base/modules/attention.py:196-219):
//   * nn.LayerNorm(dim) over the channel axis of each token (norm1 / norm2 / norm3, attention.py:204-206,215-217)
//   * GEGLU: x, gate = proj(x).chunk(2, -1); x * gelu(gate)          (attention.py:38-46; F.gelu = the exact erf form)
// Tokens are the pixels of the NHWC activation ('b c h w -> b (h w) c' is a no-op in this layout).  Both are HBM-bound
// streaming passes; the linear layers around them are 1x1 convolutions on the matrix core (csrc/conv_igemm.hip).
#include "common.h"
#include "stats_acc.h"

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

// ---- backward (training through SpatialTransformer blocks) ------------------------------------------------------------
// LayerNorm: with xh = (x - mean) rstd and g = dy gamma:  dx = rstd (g - mean(g) - xh mean(g xh)) [+ dadd: the residual
// branch's gradient, x = f(norm(x)) + x in BasicTransformerBlock._forward, attention.py:215-218];
// dgamma[c] = sum_rows dy xh, dbeta[c] = sum_rows dy.  One wavefront per token (statistics recomputed: x is re-read anyway),
// a workgroup's waves walk rows with a grid stride and add their dy xh / dy into per-WAVE LDS [4][2][C] tables (no atomics: a lane owns
// its channels), which the workgroup adds in wave order and hands to exact integer-limb cells (stats_acc.h) -- bitwise reproducible.
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, int lddy,
                                                            const float* __restrict__ dadd, int ldadd, float* __restrict__ dx,
                                                            int lddx, unsigned long long* __restrict__ ws, long long rows, int C, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lsum_all[];    // [4 waves][2][C]
    const int lane = threadIdx.x & 63;
    float* lsum = lsum_all + (size_t)(threadIdx.x >> 6) * 2 * C;       // this wave's table
    for (int i = threadIdx.x; i < 8 * C; i += 256) lsum_all[i] = 0.f;
    __syncthreads();
    const float invC = 1.0f / (float)C;
    for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
        const float* xr = x + (size_t)row * ldx;
        const float* dr = dy + (size_t)row * lddy;
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            s += (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s * invC;
        float q = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float a = v.x - mean, b = v.y - mean, d = v.z - mean, e = v.w - mean;
            q += (a * a + b * b) + (d * d + e * e);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
        const float rstd = 1.0f / sqrtf(q * invC + eps);
        float sg = 0.f, sgx = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float4 d = *reinterpret_cast<const float4*>(dr + c);
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float h0 = (v.x - mean) * rstd, h1 = (v.y - mean) * rstd, h2 = (v.z - mean) * rstd, h3 = (v.w - mean) * rstd;
            const float g0 = d.x * g.x, g1 = d.y * g.y, g2 = d.z * g.z, g3 = d.w * g.w;
            sg += (g0 + g1) + (g2 + g3);
            sgx += (g0 * h0 + g1 * h1) + (g2 * h2 + g3 * h3);
            lsum[c + 0] += d.x * h0; lsum[c + 1] += d.y * h1;           // (lane-private slots: row after row, in order)
            lsum[c + 2] += d.z * h2; lsum[c + 3] += d.w * h3;
            lsum[C + c + 0] += d.x; lsum[C + c + 1] += d.y;
            lsum[C + c + 2] += d.z; lsum[C + c + 3] += d.w;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sg += __shfl_xor(sg, off);
            sgx += __shfl_xor(sgx, off);
        }
        const float m1 = sg * invC, m2 = sgx * invC;
        float* o = dx + (size_t)row * lddx;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float4 d = *reinterpret_cast<const float4*>(dr + c);
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            float4 r;
            r.x = rstd * (d.x * g.x - m1 - (v.x - mean) * rstd * m2);
            r.y = rstd * (d.y * g.y - m1 - (v.y - mean) * rstd * m2);
            r.z = rstd * (d.z * g.z - m1 - (v.z - mean) * rstd * m2);
            r.w = rstd * (d.w * g.w - m1 - (v.w - mean) * rstd * m2);
            if (dadd) {
                const float4 e = *reinterpret_cast<const float4*>(dadd + (size_t)row * ldadd + c);
                r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w;
            }
            *reinterpret_cast<float4*>(o + c) = r;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        sa_add(ws + (size_t)i * SA_W, ((double)lsum_all[i] + (double)lsum_all[2 * C + i]) + ((double)lsum_all[4 * C + i] + (double)lsum_all[6 * C + i]));
}

__global__ void layernorm_bwd_final_kernel(const unsigned long long* __restrict__ ws, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                           int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        dgamma[c] = (float)sa_load(ws + (size_t)c * SA_W);
        dbeta[c] = (float)sa_load(ws + (size_t)(C + c) * SA_W);
    }
}

// GEGLU: y = a gelu(g)  =>  da = dy gelu(g),  dg = dy a gelu'(g),  gelu'(g) = Phi(g) + g phi(g)   (exact erf form)
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ da, int ldda, long long rows, int inner) {
    const int I4 = inner >> 2;
    const long long total = rows * I4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / I4;
        const int c = (int)(i - r * I4) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(a + (size_t)r * lda + c);
        const float4 gv = *reinterpret_cast<const float4*>(a + (size_t)r * lda + inner + c);
        const float4 dv = *reinterpret_cast<const float4*>(dy + (size_t)r * lddy + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float oa[4], og[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float cdf = 0.5f * (1.0f + erff(gs[j] * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * gs[j] * gs[j]);
            oa[j] = ds[j] * (gs[j] * cdf);
            og[j] = ds[j] * xs[j] * (cdf + gs[j] * pdf);
        }
        *reinterpret_cast<float4*>(da + (size_t)r * ldda + c) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        *reinterpret_cast<float4*>(da + (size_t)r * ldda + inner + c) = make_float4(og[0], og[1], og[2], og[3]);
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

// dx (overwritten; + dadd when given), dgamma, dbeta (overwritten) of bbdm_layernorm_f32.  ws: 2*C doubles of scratch.
extern "C" int bbdm_layernorm_bwd_f32(const float* x, int ldx, const float* gamma, const float* dy, int lddy, const float* dadd,
                                      int ldadd, float* dx, int lddx, float* dgamma, float* dbeta, double* ws_, long long rows,
                                      int C, float eps, void* stream) {
    BBDM_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && ws_ && rows > 0 && C > 0 && ((uintptr_t)ws_ & 7) == 0, "layernorm_bwd: bad args");
    unsigned long long* ws = reinterpret_cast<unsigned long long*>(ws_);         // [2][C][SA_W] limb cells (stats_acc.h)
    BBDM_REQUIRE(C % 4 == 0 && C <= 2048 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldx >= C && lddy >= C && lddx >= C &&
                     (!dadd || (ldadd % 4 == 0 && ldadd >= C)), "layernorm_bwd: C=%d ldx=%d lddy=%d lddx=%d", C, ldx, lddy, lddx);
    BBDM_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)dadd) & 15) == 0,
                 "layernorm_bwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    bbdm_zero_async(ws, 8 * (size_t)2 * C * SA_W, st);
    long long blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)blocks), dim3(256), sizeof(float) * 8 * C, st, x, ldx, gamma, dy, lddy,
                       dadd, ldadd, dx, lddx, ws, rows, C, eps);
    hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, ws, dgamma, dbeta, C);
    BBDM_CHECK_LAUNCH("layernorm_bwd");
    return BBDM_OK;
}

// da[rows][2*inner] (overwritten) of bbdm_geglu_f32 from its input a and the gradient dy[rows][inner] of its output.
extern "C" int bbdm_geglu_bwd_f32(const float* a, int lda, const float* dy, int lddy, float* da, int ldda, long long rows,
                                  int inner, void* stream) {
    BBDM_REQUIRE(a && dy && da && rows > 0 && inner > 0, "geglu_bwd: bad args");
    BBDM_REQUIRE(inner % 4 == 0 && lda % 4 == 0 && lddy % 4 == 0 && ldda % 4 == 0 && lda >= 2 * inner && ldda >= 2 * inner &&
                     lddy >= inner, "geglu_bwd: inner=%d lda=%d lddy=%d ldda=%d", inner, lda, lddy, ldda);
    BBDM_REQUIRE((((uintptr_t)a | (uintptr_t)dy | (uintptr_t)da) & 15) == 0, "geglu_bwd: 16-byte alignment");
    long long blocks = (rows * (inner / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lda, dy, lddy, da, ldda,
                       rows, inner);
    BBDM_CHECK_LAUNCH("geglu_bwd");
    return BBDM_OK;
}
