// groupnorm_bwd.hip -- backward of the fused GroupNorm -> [FiLM] -> [SiLU] -> [avg-pool | nearest x2] pass
// (training path; the reference gets it from autograd through util.py:214-216 / openaimodel.py:259-274).
//
// Forward (groupnorm.hip):  xh = (x - mu) r,  y = xh g + b,  v = y (1 + sc) + sh,  z = silu(v),  a = R(z)
// Given dA (gradient of a), with dz = R^T(dA) and dv = dz * silu'(v):
//   P[n,c] = sum_hw dv xh,   Q[n,c] = sum_hw dv                                   (kernel 1: reduce, fp64 sums)
//   dgamma[c] = sum_n (1+sc) P,  dbeta[c] = sum_n (1+sc) Q,  dsc[n,c] = g P + b Q,  dsh[n,c] = Q,
//   S1[n,G] = sum_{c in G} g (1+sc) Q,  S2[n,G] = sum_{c in G} g (1+sc) P          (kernel 2: finalize, tiny; dgamma / dbeta leave
//                                                                                    as fp32 in kernel 3's first workgroup)
//   dx = r ( g (1+sc) dv - S1/cnt - xh S2/cnt )  [+ R^T(dAdd)]  [+ dx]             (kernel 3: apply)
// dAdd carries the gradient of the skip path (x_upd / identity), which shares the same resampling.
// All three are HBM-bound streaming kernels with float4 accesses along the channel axis.
#include "common.h"
#include "stats_acc.h"

namespace {

struct BwdArgs {
    const float* x;
    const unsigned long long* stats;     // exact limb accumulators (stats_acc.h)
    const float* gamma;
    const float* beta;
    const float* film;
    const float* da;
    const float* dadd;
    const double* sg;
    const float4* coef;          // [N][G] {rstd, mean, S1/cnt, S2/cnt} in fp32, written by the finalize kernel
    double* pq;                  // [N][splits][C][2] per-workgroup partial sums of P, Q (plain stores; the finalize kernel adds the splits in order)
    const double* dgb;           // [N][C][2] per-image dgamma / dbeta terms (finalize kernel, plain stores); the apply pass adds the images in order
    int N;
    float* dgamma;
    float* dbeta;
    float* dx;
    int ldx, ldda, ldadd, lddx, film_ld;
    int H, W, C, G;
    float eps;
    int silu, resample, accumulate;
};

// silu'(v) = s (1 + v (1 - s)), s = sigmoid(v), on the hardware v_exp_f32 + v_rcp_f32 (~2 ulp each; gradient bar 1e-3): both streaming
// passes evaluate it per element, and the IEEE expf + division sequence (~25 VALU instructions) made them ALU-bound -- the same trade
// as silu_fast in the Winograd input transform (common.h)
__device__ __forceinline__ float dsilu(float v) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return s * (1.0f + v * (1.0f - s));
}

// gradient arriving at input pixel (h, w), channel quad c, from tensor `g` (pitch ld) living at the OUTPUT resolution
__device__ __forceinline__ float4 gather_grad(const float* __restrict__ g, int ld, int resample, int h, int w, int H,
                                              int W, int c) {
    if (resample == 0) return *reinterpret_cast<const float4*>(g + ((size_t)h * W + w) * ld + c);
    if (resample == 1) {
        const int Wo = W >> 1;
        float4 v = *reinterpret_cast<const float4*>(g + ((size_t)(h >> 1) * Wo + (w >> 1)) * ld + c);
        return make_float4(v.x * 0.25f, v.y * 0.25f, v.z * 0.25f, v.w * 0.25f);
    }
    if (resample == 3) {                      // forward kept pixel (2ho, 2wo): only those receive a gradient
        if ((h | w) & 1) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(g + ((size_t)(h >> 1) * (W >> 1) + (w >> 1)) * ld + c);
    }
    const int Wo = W * 2;
    const float* p0 = g + ((size_t)(2 * h) * Wo + 2 * w) * ld + c;
    const float* p1 = p0 + (size_t)Wo * ld;
    const float4 a = *reinterpret_cast<const float4*>(p0), b = *reinterpret_cast<const float4*>(p0 + ld);
    const float4 cc = *reinterpret_cast<const float4*>(p1), d = *reinterpret_cast<const float4*>(p1 + ld);
    return make_float4((a.x + b.x) + (cc.x + d.x), (a.y + b.y) + (cc.y + d.y), (a.z + b.z) + (cc.z + d.z),
                       (a.w + b.w) + (cc.w + d.w));
}

struct Norm4 {
    float rs[4], mu[4];
};

__device__ __forceinline__ Norm4 load_norm(const BwdArgs& a, int n, int c, int cpg, double cnt) {
    Norm4 o;
    const int ng = (cpg & 3) ? 4 : 1;
    for (int e = 0; e < ng; ++e) {
        const int g = (c + e) / cpg;
        const double s = sa_load(a.stats + (((size_t)n * a.G + g) * 2) * SA_W), ss = sa_load(a.stats + (((size_t)n * a.G + g) * 2 + 1) * SA_W);
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        o.rs[e] = (float)(1.0 / sqrt(var + (double)a.eps));
        o.mu[e] = (float)mean;
    }
    if (ng == 1) { o.rs[1] = o.rs[2] = o.rs[3] = o.rs[0]; o.mu[1] = o.mu[2] = o.mu[3] = o.mu[0]; }
    return o;
}

// xh (normalised input) and dv for one quad
__device__ __forceinline__ void quad_dv(const BwdArgs& a, int n, int c, const Norm4& nm, float4 xv, float4 dz,
                                        float xh[4], float dv[4]) {
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const float dzs[4] = {dz.x, dz.y, dz.z, dz.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = (xs[e] - nm.mu[e]) * nm.rs[e];
        float v = xh[e] * a.gamma[c + e] + a.beta[c + e];
        if (a.film) v = v * (1.f + a.film[(size_t)n * a.film_ld + c + e]) + a.film[(size_t)n * a.film_ld + a.C + c + e];
        dv[e] = a.silu ? dzs[e] * dsilu(v) : dzs[e];
    }
}

// ---- kernel 1: P, Q ----------------------------------------------------------------------------------------------
template <int JMAX>
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const BwdArgs a, int pix_per_block) {
    // Deterministic (round 5): the pixel rows of a workgroup meet in LDS in a FIXED order (no LDS atomics); every workgroup STORES its
    // partial sums and the finalize kernel adds the workgroups of an image in index order (no global atomics: limb cells here -- three
    // integer atomics per value and workgroup -- doubled this pass)
    __shared__ double lacc[256 * 8];          // [thread][P0..3, Q0..3]
    const int tid = threadIdx.x, n = blockIdx.y;
    const int C4 = a.C >> 2, cpg = a.C / a.G, HW = a.H * a.W;
    const double cnt = (double)HW * cpg;
#pragma unroll
    for (int e = 0; e < 8; ++e) lacc[tid * 8 + e] = 0.0;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    const int Ho = (a.resample == 1 || a.resample == 3) ? a.H >> 1 : (a.resample == 2 ? a.H * 2 : a.H);
    const int Wo = (a.resample == 1 || a.resample == 3) ? a.W >> 1 : (a.resample == 2 ? a.W * 2 : a.W);
    const float* xb = a.x + (size_t)n * HW * a.ldx;
    const float* dab = a.da + (size_t)n * Ho * Wo * a.ldda;
    int PP, prow, c4base;
    bool active;
    if (C4 <= 256) { PP = 256 / C4; prow = tid / C4; c4base = tid - prow * C4; active = prow < PP; }
    else { PP = 1; prow = 0; c4base = tid; active = true; }
    if (active) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int c4 = c4base + j * 256;
            if (c4 >= C4) continue;
            const int c = c4 * 4;
            const Norm4 nm = load_norm(a, n, c, cpg, cnt);
            double P[4] = {0, 0, 0, 0}, Q[4] = {0, 0, 0, 0};
            int p = p0 + prow;
            if (a.resample == 0) {
                // common case (no resampling): four pixels per trip, their eight loads issued before any arithmetic -- the
                // one-pixel loop below keeps a single load pair in flight per thread (latency-bound at 2.4 TB/s)
                for (; p + 3 * PP < p1; p += 4 * PP) {
                    float4 xv[4], dz[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xv[u] = *reinterpret_cast<const float4*>(xb + (size_t)(p + u * PP) * a.ldx + c);
                        dz[u] = *reinterpret_cast<const float4*>(dab + (size_t)(p + u * PP) * a.ldda + c);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float xh[4], dv[4];
                        quad_dv(a, n, c, nm, xv[u], dz[u], xh, dv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { P[e] += (double)dv[e] * xh[e]; Q[e] += (double)dv[e]; }
                    }
                }
            }
            for (; p < p1; p += PP) {
                const int h = p / a.W, w = p - h * a.W;
                const float4 xv = *reinterpret_cast<const float4*>(xb + (size_t)p * a.ldx + c);
                const float4 dz = gather_grad(dab, a.ldda, a.resample, h, w, a.H, a.W, c);
                float xh[4], dv[4];
                quad_dv(a, n, c, nm, xv, dz, xh, dv);
#pragma unroll
                for (int e = 0; e < 4; ++e) { P[e] += (double)dv[e] * xh[e]; Q[e] += (double)dv[e]; }
            }
            if (JMAX == 1 && C4 <= 256) {        // several pixel rows per channel quad: through LDS (below)
#pragma unroll
                for (int e = 0; e < 4; ++e) { lacc[tid * 8 + e] = P[e]; lacc[tid * 8 + 4 + e] = Q[e]; }
            } else {                             // one thread per channel quad: straight to this workgroup's partials
                double* o = a.pq + (((size_t)n * gridDim.x + blockIdx.x) * a.C + c) * 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[2 * e] = P[e]; o[2 * e + 1] = Q[e]; }
            }
        }
    }
    if (JMAX == 1 && C4 <= 256) {
        __syncthreads();
        for (int i = tid; i < 2 * a.C; i += 256) {               // value i = (channel, P | Q): its PP rows added in row order
            const int c = i >> 1, which = i & 1, c4 = c >> 2, e = c & 3;
            double v = 0.0;
            for (int r = 0; r < PP; ++r) v += lacc[(r * C4 + c4) * 8 + which * 4 + e];
            a.pq[((size_t)n * gridDim.x + blockIdx.x) * a.C * 2 + i] = v;
        }
    }
}

// ---- kernel 2: finalize (one block per image) --------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const double* __restrict__ pq, int splits, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ film,
                                                              int film_ld, double* __restrict__ sg, double* __restrict__ dgb,
                                                              float* __restrict__ dfilm, int dfilm_ld, int C, int G,
                                                              const unsigned long long* __restrict__ stats, float4* __restrict__ coef,
                                                              double cnt, float eps) {
    // grid = N; dgb: [N][C][2] this image's dgamma / dbeta terms (the apply pass adds the images in index order: no atomics, nothing to
    // zero); the group sums S1, S2 in LDS limb cells (integer atomics: order-independent)
    __shared__ unsigned long long s12c[64 * 2 * SA_W];
    __shared__ double s12[64 * 2];
    const int n = blockIdx.x, tid = threadIdx.x, cpg = C / G;
    for (int i = tid; i < 2 * G * SA_W; i += 256) s12c[i] = 0ull;
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double P = 0.0, Q = 0.0;                           // the pixel blocks of image n, in index order; eight loads in flight per trip
        for (int z0 = 0; z0 < splits; z0 += 8) {           // (one dependent load per trip made this tiny kernel 30 us long)
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + u < splits ? z0 + u : splits - 1;
                v[u] = *reinterpret_cast<const double2*>(pq + (((size_t)n * splits + z) * C + c) * 2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (z0 + u < splits) { P += v[u].x; Q += v[u].y; }
        }
        const double g = gamma[c], b = beta[c];
        const double one_sc = film ? 1.0 + (double)film[(size_t)n * film_ld + c] : 1.0;
        *reinterpret_cast<double2*>(dgb + ((size_t)n * C + c) * 2) = make_double2(one_sc * P, one_sc * Q);
        if (dfilm) {
            dfilm[(size_t)n * dfilm_ld + c] = (float)(g * P + b * Q);       // d scale
            dfilm[(size_t)n * dfilm_ld + C + c] = (float)Q;                 // d shift
        }
        sa_add(&s12c[(2 * (c / cpg)) * SA_W], g * one_sc * Q);
        sa_add(&s12c[(2 * (c / cpg) + 1) * SA_W], g * one_sc * P);
    }
    __syncthreads();
    if (tid < 2 * G) {
        s12[tid] = sa_load(&s12c[tid * SA_W]);
        sg[(size_t)n * G * 2 + tid] = s12[tid];
    }
    __syncthreads();
    // what the apply pass needs per (image, group), once, in fp32 -- it used to redo these fp64 divisions / square roots for
    // every channel quad of every pixel, which made an HBM-bound pass ALU-bound (2.5 TB/s)
    if (tid < G) {
        const double sm = sa_load(stats + (((size_t)n * G + tid) * 2) * SA_W), ss = sa_load(stats + (((size_t)n * G + tid) * 2 + 1) * SA_W);
        const double mean = sm / cnt;
        double var = ss / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        coef[(size_t)n * G + tid] = make_float4((float)(1.0 / sqrt(var + (double)eps)), (float)mean, (float)(s12[2 * tid] / cnt),
                                                (float)(s12[2 * tid + 1] / cnt));
    }
}

// ---- kernel 3: dx ---------------------------------------------------------------------------------------------------
// A thread keeps ONE channel quad for its whole pixel range (as the reduce pass does), so gamma / beta / FiLM / the group's
// {rstd, mean, S1/cnt, S2/cnt} are loaded once per thread instead of once per element, there is no 64-bit div / mod per element,
// and the un-resampled case has four pixels' loads in flight per trip (the one-quad-per-trip form ran at 2.4 TB/s).
struct Chan4 {
    float g[4], b[4], sc1[4], sh[4], gs[4];          // gamma, beta, 1 + FiLM scale, FiLM shift, gamma (1 + scale)
    float rs[4], mu[4], s1[4], s2[4];
};

__device__ __forceinline__ Chan4 load_chan(const BwdArgs& a, int n, int c, int cpg) {
    Chan4 k;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        k.g[e] = a.gamma[c + e];
        k.b[e] = a.beta[c + e];
        k.sc1[e] = a.film ? 1.f + a.film[(size_t)n * a.film_ld + c + e] : 1.f;
        k.sh[e] = a.film ? a.film[(size_t)n * a.film_ld + a.C + c + e] : 0.f;
        k.gs[e] = a.film ? k.g[e] * k.sc1[e] : k.g[e];
        const float4 cf = a.coef[(size_t)n * a.G + (c + e) / cpg];
        k.rs[e] = cf.x; k.mu[e] = cf.y; k.s1[e] = cf.z; k.s2[e] = cf.w;
    }
    return k;
}

__device__ __forceinline__ float4 apply_quad(const BwdArgs& a, const Chan4& k, float4 xv, float4 dz) {
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const float dzs[4] = {dz.x, dz.y, dz.z, dz.w};
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = (xs[e] - k.mu[e]) * k.rs[e];
        float v = xh * k.g[e] + k.b[e];
        if (a.film) v = v * k.sc1[e] + k.sh[e];
        const float dv = a.silu ? dzs[e] * dsilu(v) : dzs[e];
        out[e] = k.rs[e] * (k.gs[e] * dv - k.s1[e] - xh * k.s2[e]);
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

template <bool NORM>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const BwdArgs a, int pix_per_block) {
    const int n = blockIdx.y, tid = threadIdx.x;
    if (NORM && blockIdx.x == 0 && n == 0) {                     // dgamma / dbeta: the finalize kernel's per-image terms, added in image order
        for (int c = tid; c < a.C; c += 256) {
            double g = 0.0, b = 0.0;
            for (int m0 = 0; m0 < a.N; m0 += 8) {                // eight loads in flight per trip
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(a.dgb + ((size_t)min(m0 + u, a.N - 1) * a.C + c) * 2);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (m0 + u < a.N) { g += v[u].x; b += v[u].y; }
            }
            a.dgamma[c] = (float)g;
            a.dbeta[c] = (float)b;
        }
    }
    const int C4 = a.C >> 2, HW = a.H * a.W;
    const int cpg = a.G > 0 ? a.C / a.G : a.C;
    const int Ho = (a.resample == 1 || a.resample == 3) ? a.H >> 1 : (a.resample == 2 ? a.H * 2 : a.H);
    const int Wo = (a.resample == 1 || a.resample == 3) ? a.W >> 1 : (a.resample == 2 ? a.W * 2 : a.W);
    const float* xb = NORM ? a.x + (size_t)n * HW * a.ldx : nullptr;
    const float* dab = NORM ? a.da + (size_t)n * Ho * Wo * a.ldda : nullptr;
    const float* addb = a.dadd ? a.dadd + (size_t)n * Ho * Wo * a.ldadd : nullptr;
    float* dxb = a.dx + (size_t)n * HW * a.lddx;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    int PP, prow, c4base;
    if (C4 <= 256) { PP = 256 / C4; prow = tid / C4; c4base = tid - prow * C4; if (prow >= PP) return; }
    else { PP = 1; prow = 0; c4base = tid; }
    for (int c4 = c4base; c4 < C4; c4 += 256) {
        const int c = c4 * 4;
        Chan4 k;
        if (NORM) k = load_chan(a, n, c, cpg);
        int p = p0 + prow;
        if (a.resample == 0) {
            for (; p + 3 * PP < p1; p += 4 * PP) {
                float4 xv[4], dz[4], ad[4], old[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t q = (size_t)(p + u * PP);
                    if (NORM) {
                        xv[u] = *reinterpret_cast<const float4*>(xb + q * a.ldx + c);
                        dz[u] = *reinterpret_cast<const float4*>(dab + q * a.ldda + c);
                    }
                    if (addb) ad[u] = *reinterpret_cast<const float4*>(addb + q * a.ldadd + c);
                    if (a.accumulate) old[u] = *reinterpret_cast<const float4*>(dxb + q * a.lddx + c);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 o = NORM ? apply_quad(a, k, xv[u], dz[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (addb) { o.x += ad[u].x; o.y += ad[u].y; o.z += ad[u].z; o.w += ad[u].w; }
                    if (a.accumulate) { o.x += old[u].x; o.y += old[u].y; o.z += old[u].z; o.w += old[u].w; }
                    *reinterpret_cast<float4*>(dxb + (size_t)(p + u * PP) * a.lddx + c) = o;
                }
            }
        }
        for (; p < p1; p += PP) {
            const int h = p / a.W, w = p - h * a.W;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NORM) {
                const float4 xv = *reinterpret_cast<const float4*>(xb + (size_t)p * a.ldx + c);
                const float4 dz = gather_grad(dab, a.ldda, a.resample, h, w, a.H, a.W, c);
                o = apply_quad(a, k, xv, dz);
            }
            if (addb) {
                const float4 ad = gather_grad(addb, a.ldadd, a.resample, h, w, a.H, a.W, c);
                o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
            }
            float* op = dxb + (size_t)p * a.lddx + c;
            if (a.accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(op);
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *reinterpret_cast<float4*>(op) = o;
        }
    }
}

}  // namespace

// pixel blocks per image of the reduce pass (its grid.x; upper bound used by the workspace query)
static int gn_bwd_splits(int N, int HW, int C) {
    const int C4 = C / 4, PP = C4 <= 256 ? 256 / C4 : 1;
    int splits = cdiv(1024, N);
    int ppb = cdiv(HW, splits);
    if (ppb < PP * 8) ppb = PP * 8;
    return cdiv(HW, ppb);
}
// Workspace (8-byte elements): pq [N][splits <= ceil(1024 / N)][C][2] fp64 partials + dgb [N][C][2] fp64 + sg [N][G][2] fp64 +
// coef [N][G] float4 (= 2 doubles each).  Nothing in it needs zeroing (every element read is stored first).
extern "C" size_t bbdm_groupnorm_bwd_workspace_doubles(int N, int C, int G) {
    return (size_t)N * cdiv(1024, N > 0 ? N : 1) * C * 2 + (size_t)N * C * 2 + (size_t)N * G * 2 + (size_t)N * G * 2;
}

extern "C" int bbdm_groupnorm_bwd_f32(const float* x, int ldx, const void* stats_, const float* gamma, const float* beta,
                                      const float* film, int film_ld, const float* da, int ldda, const float* dadd,
                                      int ldadd, float* dx, int lddx, int accumulate, float* dgamma, float* dbeta,
                                      float* dfilm, int dfilm_ld, double* ws, int N, int H, int W, int C, int G, float eps,
                                      int silu, int resample, void* stream) {
    BBDM_REQUIRE(dx, "gn_bwd: null dx");
    const int norm = gamma != nullptr;
    BBDM_REQUIRE(norm || dadd, "gn_bwd: nothing to do (no norm, no dadd)");
    const unsigned long long* stats = (const unsigned long long*)stats_;
    BBDM_REQUIRE(!norm || (x && stats && beta && da && dgamma && dbeta && ws), "gn_bwd: missing pointer for the norm path");
    BBDM_REQUIRE(resample >= 0 && resample <= 3 && ((resample != 1 && resample != 3) || (H % 2 == 0 && W % 2 == 0)),
                 "gn_bwd: resample");
    BBDM_REQUIRE(N > 0 && H > 0 && W > 0 && C % 4 == 0 && lddx % 4 == 0 && lddx >= C, "gn_bwd: shape/pitch");
    BBDM_REQUIRE(!norm || (ldx % 4 == 0 && ldda % 4 == 0 && G > 0 && G <= 64 && C % G == 0 && C <= 4096), "gn_bwd: norm args");
    BBDM_REQUIRE(!dadd || ldadd % 4 == 0, "gn_bwd: ldadd");
    BBDM_REQUIRE(!film || film_ld % 4 == 0, "gn_bwd: film_ld");
    hipStream_t st = (hipStream_t)stream;
    BwdArgs a;
    a.x = x; a.stats = stats; a.gamma = gamma; a.beta = beta; a.film = film; a.da = da; a.dadd = dadd; a.dx = dx;
    a.ldx = ldx; a.ldda = ldda; a.ldadd = ldadd; a.lddx = lddx; a.film_ld = film_ld;
    a.H = H; a.W = W; a.C = C; a.G = norm ? G : 1; a.eps = eps; a.silu = silu; a.resample = resample;
    a.N = N;
    a.accumulate = accumulate; a.pq = nullptr; a.sg = nullptr; a.coef = nullptr; a.dgb = nullptr; a.dgamma = nullptr; a.dbeta = nullptr;
    const int HW = H * W;
    if (norm) {
        const int nsplit = gn_bwd_splits(N, HW, C);
        double* pq = ws;
        double* dgb = pq + (size_t)N * nsplit * C * 2;
        double* sg = dgb + (size_t)N * C * 2;
        float4* coef = reinterpret_cast<float4*>(sg + (size_t)N * G * 2);
        a.pq = pq; a.sg = sg; a.coef = coef;
        const int C4 = C / 4;
        const int PP = C4 <= 256 ? 256 / C4 : 1;
        int splits = cdiv(1024, N);
        int ppb = cdiv(HW, splits);
        if (ppb < PP * 8) ppb = PP * 8;
        splits = cdiv(HW, ppb);
        const dim3 grid(splits, N);
        if (C4 <= 256) hipLaunchKernelGGL(gn_bwd_reduce_kernel<1>, grid, dim3(256), 0, st, a, ppb);
        else if (C4 <= 512) hipLaunchKernelGGL(gn_bwd_reduce_kernel<2>, grid, dim3(256), 0, st, a, ppb);
        else hipLaunchKernelGGL(gn_bwd_reduce_kernel<4>, grid, dim3(256), 0, st, a, ppb);
        hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N), dim3(256), 0, st, pq, splits, gamma, beta, film, film_ld, sg, dgb, dfilm,
                           dfilm_ld, C, G, stats, coef, (double)HW * (C / G), eps);
        a.dgb = dgb; a.dgamma = dgamma; a.dbeta = dbeta;
    }
    {
        const int C4 = C / 4;
        const int PP = C4 <= 256 ? 256 / C4 : 1;
        int splits = cdiv(4096, N);                              // ~16 workgroups per CU over the batch
        int ppb = cdiv(HW, splits);
        if (ppb < PP * 4) ppb = PP * 4;
        ppb = cdiv(ppb, PP * 4) * (PP * 4);                      // whole four-pixel trips
        splits = cdiv(HW, ppb);
        if (norm) hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3((unsigned)splits, N), dim3(256), 0, st, a, ppb);
        else hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3((unsigned)splits, N), dim3(256), 0, st, a, ppb);
    }
    BBDM_CHECK_LAUNCH("gn_bwd");
    return BBDM_OK;
}
