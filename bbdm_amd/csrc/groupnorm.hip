// groupnorm.hip -- GroupNorm(32 groups) on fp32 NHWC tensors, split into
//   (1) a statistics pass (per (n, group) sum / sum-of-squares, accumulated in fp64) and
//   (2) a fused apply pass:  GN affine -> [FiLM (1+scale), shift] -> [SiLU] -> [2x2 avg-pool | nearest x2].
//
// Replaces util.py:199-216 (GroupNorm32) at openaimodel.py:205,229,306,688, the FiLM arithmetic at
// openaimodel.py:270-273, nn.SiLU, Downsample/Upsample without conv (openaimodel.py:118,159) at :259-264.
//
// Both passes are HBM-bound streaming kernels: 16-byte (float4) accesses along the contiguous channel
// axis, one wave covers 1 KiB of consecutive bytes, many small blocks so that all 256 CUs stream.
#include "common.h"
#include "stats_acc.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// stats: grid = (pixel splits, N).  Thread t owns channel-quads c4 = t % C4 (+ k*256 when C4 > 256) and walks
// pixels with stride PP = 256 / C4; its accumulators therefore always belong to one group per quad.
// ---------------------------------------------------------------------------------------------------------
template <int JMAX>
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, int ldx, unsigned long long* __restrict__ stats,
                                                       int HW, int C, int G, int pix_per_block) {
    __shared__ unsigned long long lsum[64 * 2 * SA_W];      // G <= 64; exact limb accumulators (stats_acc.h)
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int C4 = C >> 2;
    const int cpg = C / G;
    for (int i = tid; i < 2 * G * SA_W; i += 256) lsum[i] = 0ull;
    __syncthreads();

    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    const float* xb = x + (size_t)n * HW * ldx;

    int PP, prow, c4base;
    bool active;
    if (C4 <= 256) {
        PP = 256 / C4;
        prow = tid / C4;
        c4base = tid - prow * C4;
        active = prow < PP;
    } else {
        PP = 1;
        prow = 0;
        c4base = tid;
        active = true;
    }
    double s[JMAX], ss[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) s[j] = ss[j] = 0.0;
    if (active) {
        for (int p = p0 + prow; p < p1; p += PP) {
            const float* row = xb + (size_t)p * ldx;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int c4 = c4base + j * 256;
                if (c4 < C4) {
                    const float4 v = *reinterpret_cast<const float4*>(row + c4 * 4);
                    s[j] += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
                    ss[j] += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int c4 = c4base + j * 256;
            if (c4 < C4) {
                const int g = (c4 * 4) / cpg;     // cpg % 4 == 0 -> a quad never straddles two groups
                sa_add(lsum + (size_t)(2 * g) * SA_W, s[j]);
                sa_add(lsum + (size_t)(2 * g + 1) * SA_W, ss[j]);
            }
        }
    }
    __syncthreads();
    if (tid < 2 * G) sa_add_cell(stats + ((size_t)n * G * 2 + tid) * SA_W, lsum + (size_t)tid * SA_W);
}

// Scalar variant for channels-per-group not a multiple of 4 (e.g. C = 32, 96, 192 with 32 groups: only the tiny
// test configurations; every reference template has cpg % 4 == 0).  One thread per element, LDS limb cells (integer atomics) per group.
__global__ void __launch_bounds__(256) gn_stats_scalar_kernel(const float* __restrict__ x, int ldx,
                                                              unsigned long long* __restrict__ stats, int HW, int C, int G,
                                                              int pix_per_block) {
    __shared__ unsigned long long lsum[64 * 2 * SA_W];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int cpg = C / G;
    for (int i = tid; i < 2 * G * SA_W; i += 256) lsum[i] = 0ull;
    __syncthreads();
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    const float* xb = x + (size_t)n * HW * ldx;
    const long long total = (long long)(p1 - p0) * C;
    for (long long i = tid; i < total; i += 256) {
        const int p = p0 + (int)(i / C), c = (int)(i % C);
        const double v = xb[(size_t)p * ldx + c];
        sa_add(lsum + (size_t)(2 * (c / cpg)) * SA_W, v);
        sa_add(lsum + (size_t)(2 * (c / cpg) + 1) * SA_W, v * v);
    }
    __syncthreads();
    if (tid < 2 * G) sa_add_cell(stats + ((size_t)n * G * 2 + tid) * SA_W, lsum + (size_t)tid * SA_W);
}

// ---------------------------------------------------------------------------------------------------------
// apply
// ---------------------------------------------------------------------------------------------------------
struct ApplyArgs {
    const float* x;
    const unsigned long long* stats;     // exact limb accumulators (stats_acc.h): sa_load folds a cell into its fp64 sum
    const float* gamma;
    const float* beta;
    const float* film;
    float* y;
    int ldx, ldy, film_ld;
    int H, W, C, G;
    float eps;
    int silu, resample, norm;
};

__device__ __forceinline__ float4 gn_xform(const ApplyArgs& a, float4 v, float4 sc, float4 bi) {
    if (a.norm) {
        v.x = v.x * sc.x + bi.x;
        v.y = v.y * sc.y + bi.y;
        v.z = v.z * sc.z + bi.z;
        v.w = v.w * sc.w + bi.w;
    }
    if (a.silu) {
        v.x = silu_f(v.x);
        v.y = silu_f(v.y);
        v.z = silu_f(v.z);
        v.w = silu_f(v.w);
    }
    return v;
}

// grid = (blocks over out-units, N).  A "unit" is one float4 of one INPUT pixel (resample 0, 2) or of one OUTPUT
// pixel (resample 1).
__global__ void __launch_bounds__(256) gn_apply_kernel(const ApplyArgs a) {
    const int n = blockIdx.y;
    const int C4 = a.C >> 2;
    const bool halve = a.resample == 1 || a.resample == 3 || a.resample == 4;
    const int Hu = halve ? a.H >> 1 : a.H;
    const int Wu = halve ? a.W >> 1 : a.W;
    const long long units = (long long)Hu * Wu * C4;
    const int cpg = a.C / a.G;
    const double cnt = (double)a.H * a.W * cpg;
    const int Ho = halve ? a.H >> 1 : (a.resample == 2 ? a.H * 2 : a.H);
    const int Wo = halve ? a.W >> 1 : (a.resample == 2 ? a.W * 2 : a.W);
    const float* xb = a.x + (size_t)n * a.H * a.W * a.ldx;
    float* yb = a.y + (size_t)n * Ho * Wo * a.ldy;

    // x * sc + bi of channel quad c (fp64 mean / rstd per group: ~100 instructions).  The launcher sizes the grid so that its stride is a
    // multiple of C / 4 wherever it can: the quad of a thread is then the same for all its units and the coefficients are computed ONCE
    // per thread instead of once per float4 (the pass is HBM-bound only without that arithmetic).
    auto coeffs = [&](int c, float4& sc, float4& bi) {
        sc = make_float4(1.f, 1.f, 1.f, 1.f);
        bi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.norm) {
            float rs[4], mu[4];
            const int ng = (cpg & 3) ? 4 : 1;      // a quad may span several groups unless cpg % 4 == 0
            for (int e = 0; e < ng; ++e) {
                const int g = (c + e) / cpg;
                const double s = sa_load(a.stats + (((size_t)n * a.G + g) * 2) * SA_W), ss = sa_load(a.stats + (((size_t)n * a.G + g) * 2 + 1) * SA_W);
                const double mean = s / cnt;
                double var = ss / cnt - mean * mean;
                var = var > 0.0 ? var : 0.0;
                rs[e] = (float)(1.0 / sqrt(var + (double)a.eps));
                mu[e] = (float)mean;
            }
            if (ng == 1) { rs[1] = rs[2] = rs[3] = rs[0]; mu[1] = mu[2] = mu[3] = mu[0]; }
            const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c);
            const float4 be = *reinterpret_cast<const float4*>(a.beta + c);
            // y = (x - mean) * rstd * gamma + beta  ==  x * sc + bi
            sc = make_float4(rs[0] * ga.x, rs[1] * ga.y, rs[2] * ga.z, rs[3] * ga.w);
            bi = make_float4(be.x - mu[0] * sc.x, be.y - mu[1] * sc.y, be.z - mu[2] * sc.z, be.w - mu[3] * sc.w);
            if (a.film) {
                // h = GN(h) * (1 + scale) + shift     (openaimodel.py:272-273)
                const float4 fs = *reinterpret_cast<const float4*>(a.film + (size_t)n * a.film_ld + c);
                const float4 fb = *reinterpret_cast<const float4*>(a.film + (size_t)n * a.film_ld + a.C + c);
                sc = make_float4(sc.x * (1.f + fs.x), sc.y * (1.f + fs.y), sc.z * (1.f + fs.z), sc.w * (1.f + fs.w));
                bi = make_float4(bi.x * (1.f + fs.x) + fb.x, bi.y * (1.f + fs.y) + fb.y, bi.z * (1.f + fs.z) + fb.z,
                                 bi.w * (1.f + fs.w) + fb.w);
            }
        }
    };
    const long long stride = (long long)gridDim.x * 256;
    const bool fixed_quad = stride % C4 == 0;
    float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), bi0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fixed_quad) coeffs((int)((blockIdx.x * 256ll + threadIdx.x) % C4) * 4, sc0, bi0);
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < units; u += stride) {
        const int c4 = (int)(u % C4);
        const int pu = (int)(u / C4);
        const int c = c4 * 4;
        float4 sc = sc0, bi = bi0;
        if (!fixed_quad) coeffs(c, sc, bi);
        if (a.resample == 0) {
            const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)pu * a.ldx + c);
            *reinterpret_cast<float4*>(yb + (size_t)pu * a.ldy + c) = gn_xform(a, v, sc, bi);
        } else if (a.resample == 3 || a.resample == 4) {     // keep the even (3) / odd (4) positions of both axes
            const int ho = pu / Wu, wo = pu - ho * Wu, o = a.resample == 4 ? 1 : 0;
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)(2 * ho + o) * a.W + 2 * wo + o) * a.ldx + c);
            *reinterpret_cast<float4*>(yb + (size_t)pu * a.ldy + c) = gn_xform(a, v, sc, bi);
        } else if (a.resample == 1) {
            const int ho = pu / Wu, wo = pu - ho * Wu;
            const float* r0 = xb + ((size_t)(2 * ho) * a.W + 2 * wo) * a.ldx + c;
            const float* r1 = r0 + (size_t)a.W * a.ldx;
            const float4 v00 = gn_xform(a, *reinterpret_cast<const float4*>(r0), sc, bi);
            const float4 v01 = gn_xform(a, *reinterpret_cast<const float4*>(r0 + a.ldx), sc, bi);
            const float4 v10 = gn_xform(a, *reinterpret_cast<const float4*>(r1), sc, bi);
            const float4 v11 = gn_xform(a, *reinterpret_cast<const float4*>(r1 + a.ldx), sc, bi);
            float4 o;
            o.x = (((v00.x + v01.x) + v10.x) + v11.x) * 0.25f;
            o.y = (((v00.y + v01.y) + v10.y) + v11.y) * 0.25f;
            o.z = (((v00.z + v01.z) + v10.z) + v11.z) * 0.25f;
            o.w = (((v00.w + v01.w) + v10.w) + v11.w) * 0.25f;
            *reinterpret_cast<float4*>(yb + (size_t)pu * a.ldy + c) = o;
        } else {
            const int h = pu / a.W, w = pu - h * a.W;
            const float4 v = gn_xform(a, *reinterpret_cast<const float4*>(xb + (size_t)pu * a.ldx + c), sc, bi);
            float* o0 = yb + ((size_t)(2 * h) * Wo + 2 * w) * a.ldy + c;
            float* o1 = o0 + (size_t)Wo * a.ldy;
            *reinterpret_cast<float4*>(o0) = v;
            *reinterpret_cast<float4*>(o0 + a.ldy) = v;
            *reinterpret_cast<float4*>(o1) = v;
            *reinterpret_cast<float4*>(o1 + a.ldy) = v;
        }
    }
}

// Per-image, per-channel coefficients of the fused apply:  GN(x)[*(1+scale)+shift] == x * sc[n][c] + bi[n][c].
// Same expressions (and therefore the same roundings) as gn_apply_kernel.
__global__ void gn_coeffs_kernel(const unsigned long long* __restrict__ stats, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ film, int film_ld,
                                 float* __restrict__ sc_out, float* __restrict__ bi_out, int ld, int N, int HW, int C, int G,
                                 float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const int cpg = C / G, g = c / cpg;
    const double cnt = (double)HW * cpg;
    const double s = sa_load(stats + (((size_t)n * G + g) * 2) * SA_W), ss = sa_load(stats + (((size_t)n * G + g) * 2 + 1) * SA_W);
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    float sc = rstd * gamma[c];
    float bi = beta[c] - fmean * sc;
    if (film) {
        const float fs = film[(size_t)n * film_ld + c], fb = film[(size_t)n * film_ld + C + c];
        sc = sc * (1.f + fs);
        bi = bi * (1.f + fs) + fb;
    }
    sc_out[(size_t)n * ld + c] = sc;
    bi_out[(size_t)n * ld + c] = bi;
}

}  // namespace

extern "C" int bbdm_groupnorm_coeffs_f32(const void* stats, const float* gamma, const float* beta, const float* film,
                                         int film_ld, float* scale_out, float* bias_out, int ld, int N, int HW, int C, int G,
                                         float eps, void* stream) {
    BBDM_REQUIRE(stats && gamma && beta && scale_out && bias_out, "gn_coeffs: null pointer");
    BBDM_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && ld >= C, "gn_coeffs: bad shape");
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)stats, gamma, beta, film,
                       film_ld, scale_out, bias_out, ld, N, HW, C, G, eps);
    BBDM_CHECK_LAUNCH("gn_coeffs");
    return BBDM_OK;
}

// ---- bounds for the fp16-pair planes (round 6; h2_split.h) -------------------------------------------------------------------------
// A convolution input that is GroupNorm -> [FiLM] -> [SiLU] of some tensor is bounded WITHOUT looking at the tensor: a z-score over n_g
// values cannot exceed sqrt(n_g - 1), so |GN(x)[n,c] (1 + s) + t| <= |gamma_c (1 + s_nc)| sqrt(n_g - 1) + |beta_c (1 + s_nc) + t_nc|
// and |SiLU(v)| <= |v|.  One launch per forward evaluates that for EVERY such layer of the plan (one workgroup per layer: the maximum over
// (n, c) needs no atomics) and multiplies by the gain of the consumer's transform: bounds[layer] >= max |V| of that layer.
namespace {
struct H2GnLayer {            // 32 bytes (bbdm_amd/unet.py packs it with struct 'QQiiff')
    const float* gamma;       // [C]
    const float* beta;        // [C]
    int film_off;             // FiLM scale at film[n][film_off + c], shift at film[n][film_off + C + c]; < 0: none
    int C;
    float zmax;               // sqrt(values per (image, group) - 1)
    float gain;               // the consumer's transform: |B^T d B| <= gain max |d| (winograd_math.h: wino_input_gain); 1 for a 1x1 conv
};
__global__ void __launch_bounds__(256) h2_gn_bounds_kernel(const H2GnLayer* __restrict__ table, const float* __restrict__ film, int film_ld,
                                                           int N, float* __restrict__ bounds) {
    const H2GnLayer L = table[blockIdx.x];
    float m = 0.f;
    for (int i = threadIdx.x; i < N * L.C; i += 256) {
        const int n = i / L.C, c = i - n * L.C;
        float a = L.gamma[c], b = L.beta[c];
        if (L.film_off >= 0) {
            const float fs = film[(size_t)n * film_ld + L.film_off + c], fb = film[(size_t)n * film_ld + L.film_off + L.C + c];
            a = a * (1.f + fs);
            b = b * (1.f + fs) + fb;
        }
        m = fmaxf(m, fabsf(a) * L.zmax + fabsf(b));
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) bounds[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * L.gain;
}
// A tensor whose GroupNorm statistics exist is bounded by them: |x_i| <= sqrt(sum_j x_j^2) over its (image, group).  bound = the largest
// such root over the N x G accumulator cells (exact sums, stats_acc.h) -- loose by up to sqrt(n_g), which the fp16 pair's 18 binades of
// full precision absorb (h2_split.h) -- for the 1x1 convolutions that read the RAW block input (the skip projections of the ResBlocks).
__global__ void __launch_bounds__(256) h2_stats_bound_kernel(const unsigned long long* __restrict__ stats, int cells, float* __restrict__ bound) {
    float m = 0.f;
    for (int i = threadIdx.x; i < cells; i += 256) {
        const double ss = sa_load(stats + ((size_t)i * 2 + 1) * SA_W);
        m = fmaxf(m, (float)(sqrt(ss > 0.0 ? ss : 0.0) * 1.000001));
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *bound = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// The output of a 1x1 convolution / Linear y = W x + b is bounded by max_row sum_k |W[row][k]| x max |x| + max |b|: the first and the
// last factor are properties of the weights (h2_rowl1_kernel, at packing time), the middle one is the input's bound of every forward
// (h2_affine_bound_kernel: one thread) -- the bound behind the attention's fp16-pair planes (csrc/attention.hip), whose operands are
// the qkv projection of a GroupNorm output.
__global__ void h2_zero2_kernel(float* __restrict__ out2) { out2[0] = 0.f; out2[1] = 0.f; }
__global__ void __launch_bounds__(256) h2_rowl1_kernel(const float* __restrict__ w, const float* __restrict__ bias, int rows, int cols,
                                                       float* __restrict__ out2) {
    // a wave per row, lanes along the row; one atomic maximum per workgroup into the zeroed pair (non-negative floats order like their
    // bits; a maximum does not depend on the order of the atomics).  Training plans run this after every optimizer step: as ONE
    // workgroup it took 3.8 ms for the 3072 x 1024 qkv weight of the LBBDM-f4 UNet (profiles/r06_c3_kernel_stats.md of that build)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float l1 = 0.f, bm = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float a = 0.f;
        for (int k = lane; k < cols; k += 64) a += fabsf(w[(size_t)row * cols + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        l1 = fmaxf(l1, a);
    }
    if (bias && blockIdx.x == 0)
        for (int k = threadIdx.x; k < rows; k += 256) bm = fmaxf(bm, fabsf(bias[k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o));
    __shared__ float red[4], redb[4];
    if (lane == 0) { red[wave] = l1; redb[wave] = bm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (sums of non-negative fp32 terms: every rounding is below 2^-24 relative; the margin covers rows of 2^16 elements)
        atomicMax(reinterpret_cast<unsigned*>(out2), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * 1.004f));
        if (blockIdx.x == 0) out2[1] = fmaxf(fmaxf(redb[0], redb[1]), fmaxf(redb[2], redb[3]));
    }
}
__global__ void h2_affine_bound_kernel(const float* __restrict__ in, const float* __restrict__ gain2, float* __restrict__ out) {
    *out = *in * gain2[0] * 1.00001f + gain2[1];
}
}  // namespace

extern "C" int bbdm_h2_rowl1_f32(const float* w, const float* bias, int rows, int cols, float* out2, void* stream) {
    BBDM_REQUIRE(w && out2 && rows > 0 && cols > 0, "h2_rowl1: bad args");
    hipLaunchKernelGGL(h2_zero2_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out2);
    hipLaunchKernelGGL(h2_rowl1_kernel, dim3((unsigned)min(256, cdiv(rows, 4))), dim3(256), 0, (hipStream_t)stream, w, bias, rows, cols, out2);
    BBDM_CHECK_LAUNCH("h2_rowl1");
    return BBDM_OK;
}
extern "C" int bbdm_h2_affine_bound_f32(const float* in_bound, const float* gain2, float* out_bound, void* stream) {
    BBDM_REQUIRE(in_bound && gain2 && out_bound, "h2_affine_bound: null pointer");
    hipLaunchKernelGGL(h2_affine_bound_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, in_bound, gain2, out_bound);
    BBDM_CHECK_LAUNCH("h2_affine_bound");
    return BBDM_OK;
}

extern "C" int bbdm_h2_stats_bound_f32(const void* stats, int N, int G, float* bound, void* stream) {
    BBDM_REQUIRE(stats && bound && N > 0 && G > 0, "h2_stats_bound: bad args");
    hipLaunchKernelGGL(h2_stats_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)stats, N * G, bound);
    BBDM_CHECK_LAUNCH("h2_stats_bound");
    return BBDM_OK;
}

extern "C" int bbdm_h2_gn_bounds_f32(const void* table, int nlayers, const float* film, int film_ld, int N, float* bounds, void* stream) {
    BBDM_REQUIRE(table && bounds && nlayers > 0 && N > 0, "h2_gn_bounds: bad args");
    hipLaunchKernelGGL(h2_gn_bounds_kernel, dim3(nlayers), dim3(256), 0, (hipStream_t)stream, (const H2GnLayer*)table, film, film_ld, N,
                       bounds);
    BBDM_CHECK_LAUNCH("h2_gn_bounds");
    return BBDM_OK;
}

// ---- the accumulator itself (layout: stats_acc.h) -------------------------------------------------------------------------------
extern "C" size_t bbdm_groupnorm_stats_bytes(int N, int G) {
    return N > 0 && G > 0 ? (size_t)N * G * 2 * SA_W * sizeof(unsigned long long) : 0;
}
namespace {
__global__ void gn_stats_read_kernel(const unsigned long long* __restrict__ acc, double* __restrict__ out, int cells) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells) out[i] = sa_load(acc + (size_t)i * SA_W);
}
}  // namespace
extern "C" int bbdm_groupnorm_stats_read_f64(const void* stats, double* sums_out, int N, int G, void* stream) {
    BBDM_REQUIRE(stats && sums_out && N > 0 && G > 0, "gn_stats_read: bad args");
    const int cells = N * G * 2;
    hipLaunchKernelGGL(gn_stats_read_kernel, dim3(cdiv(cells, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)stats, sums_out, cells);
    BBDM_CHECK_LAUNCH("gn_stats_read");
    return BBDM_OK;
}

extern "C" int bbdm_groupnorm_stats_f32(const float* x, int ldx, void* stats_, int N, int HW, int C, int G,
                                        void* stream) {
    unsigned long long* stats = (unsigned long long*)stats_;
    BBDM_REQUIRE(x && stats, "gn_stats: null pointer");
    BBDM_REQUIRE(N > 0 && HW > 0 && G > 0 && G <= 64 && C % G == 0, "gn_stats: bad shape C=%d G=%d", C, G);
    BBDM_REQUIRE(ldx >= C, "gn_stats: ldx < C");
    if ((C / G) % 4 != 0) {
        int splits = cdiv(1024, N);
        int ppb = cdiv(HW, splits);
        if (ppb < 16) ppb = 16;
        hipLaunchKernelGGL(gn_stats_scalar_kernel, dim3(cdiv(HW, ppb), N), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           stats, HW, C, G, ppb);
        BBDM_CHECK_LAUNCH("gn_stats");
        return BBDM_OK;
    }
    BBDM_REQUIRE(ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "gn_stats: ldx (%d) must be a multiple of 4, x 16-B aligned",
                 ldx);
    const int C4 = C / 4;
    BBDM_REQUIRE(C4 <= 1024, "gn_stats: C=%d > 4096 unsupported", C);
    // ~2048 blocks in total, but at least 8 pixel-rows of work per thread row
    const int PP = C4 <= 256 ? 256 / C4 : 1;
    int splits = cdiv(2048, N);
    int ppb = cdiv(HW, splits);
    if (ppb < PP * 8) ppb = PP * 8;
    splits = cdiv(HW, ppb);
    dim3 grid(splits, N);
    hipStream_t st = (hipStream_t)stream;
    if (C4 <= 256)
        hipLaunchKernelGGL(gn_stats_kernel<1>, grid, dim3(256), 0, st, x, ldx, stats, HW, C, G, ppb);
    else if (C4 <= 512)
        hipLaunchKernelGGL(gn_stats_kernel<2>, grid, dim3(256), 0, st, x, ldx, stats, HW, C, G, ppb);
    else
        hipLaunchKernelGGL(gn_stats_kernel<4>, grid, dim3(256), 0, st, x, ldx, stats, HW, C, G, ppb);
    BBDM_CHECK_LAUNCH("gn_stats");
    return BBDM_OK;
}

extern "C" int bbdm_groupnorm_apply_f32(const float* x, int ldx, const void* stats, const float* gamma,
                                        const float* beta, const float* film, int film_ld, float* y, int ldy, int N,
                                        int H, int W, int C, int G, float eps, int silu, int resample, void* stream) {
    BBDM_REQUIRE(x && y, "gn_apply: null pointer");
    const int norm = gamma != nullptr;
    BBDM_REQUIRE(!norm || (stats && beta), "gn_apply: gamma given without stats/beta");
    BBDM_REQUIRE(norm || !film, "gn_apply: film without norm");
    BBDM_REQUIRE(resample >= 0 && resample <= 4, "gn_apply: resample=%d", resample);
    BBDM_REQUIRE((resample != 1 && resample < 3) || (H % 2 == 0 && W % 2 == 0), "gn_apply: down-sampling needs even H, W");
    BBDM_REQUIRE(N > 0 && H > 0 && W > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C,
                 "gn_apply: bad shape / pitch");
    BBDM_REQUIRE(!norm || (G > 0 && C % G == 0), "gn_apply: C %% G != 0");
    BBDM_REQUIRE(!film || film_ld % 4 == 0, "gn_apply: film_ld %% 4");
    BBDM_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "gn_apply: x / y must be 16-byte aligned");
    ApplyArgs a;
    a.x = x; a.stats = (const unsigned long long*)stats; a.gamma = gamma; a.beta = beta; a.film = film; a.y = y;
    a.ldx = ldx; a.ldy = ldy; a.film_ld = film_ld;
    a.H = H; a.W = W; a.C = C; a.G = norm ? G : 1; a.eps = eps; a.silu = silu; a.resample = resample; a.norm = norm;
    const long long units = (long long)((resample == 1 || resample >= 3) ? (H / 2) * (W / 2) : H * W) * (C / 4);
    long long blocks = (units + 255) / 256;
    const long long cap = cdiv(8192, N) > 1 ? cdiv(8192, N) : 1;
    if (blocks > cap) blocks = cap;
    {   // grid stride = a multiple of C / 4: every thread keeps one channel quad (gn_apply_kernel computes its coefficients once)
        int g = 256, r = C / 4;
        while (r) { const int t = g % r; g = r; r = t; }
        const int m = (C / 4) / g;
        if (blocks >= m) blocks = blocks / m * m;
    }
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, a);
    BBDM_CHECK_LAUNCH("gn_apply");
    return BBDM_OK;
}
