// bridge.hip -- Brownian-Bridge scheduler arithmetic and tensor-layout glue (HBM-bound elementwise kernels).
//
// Replaces q_sample (BBM.py:128-146), predict_x0_from_objective (:148-160), the p_sample update (:186-201), the
// L1/L2 loss reduction (:114-117), extract() (model/utils.py:4-7, folded in: kernels read the schedule tables
// directly), th.cat([x, context], 1) (openaimodel.py:742) and the NCHW<->NHWC hand-over at the UNet boundary.
// Compiled with -ffp-contract=off: the reference evaluates these formulas as separate fp32 tensor ops.
#include "common.h"
#include "stats_acc.h"

namespace {

__device__ __forceinline__ float predict_x0_one(int objective, float x_t, float y, float pred, float m, float sig) {
    if (objective == 0) return x_t - pred;
    if (objective == 1) return (x_t - m * y - sig * pred) / (1.f - m);
    return y - pred;
}

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ y,
                                const float* __restrict__ noise, const int64_t* __restrict__ t,
                                const float* __restrict__ m_t, const float* __restrict__ var_t,
                                float* __restrict__ x_t, float* __restrict__ target, int per_sample, size_t total,
                                int objective) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / per_sample);
        const int64_t tt = t[n];
        const float m = m_t[tt];
        const float sig = sqrtf(var_t[tt]);
        const float a = x0[i], b = y[i], e = noise[i];
        float tg;
        if (objective == 0) tg = m * (b - a) + sig * e;
        else if (objective == 1) tg = e;
        else tg = b - a;
        x_t[i] = (1.f - m) * a + m * b + sig * e;
        target[i] = tg;
    }
}

__global__ void predict_x0_kernel(const float* __restrict__ x_t, const float* __restrict__ y,
                                  const float* __restrict__ pred, const int64_t* __restrict__ t,
                                  const float* __restrict__ m_t, const float* __restrict__ var_t,
                                  float* __restrict__ x0r, int per_sample, size_t total, int objective) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / per_sample);
        const int64_t tt = t[n];
        x0r[i] = predict_x0_one(objective, x_t[i], y[i], pred[i], m_t[tt], sqrtf(var_t[tt]));
    }
}

__global__ void p_step_kernel(const float* __restrict__ x_t, const float* __restrict__ y, const float* __restrict__ pred,
                              const float* __restrict__ noise, const float* __restrict__ m_tab,
                              const float* __restrict__ var_tab, int t, int t_next, int is_last, float eta, int clip,
                              int objective, float* __restrict__ x_next, float* __restrict__ x0_recon,
                              float* __restrict__ x_next_alias, size_t total) {
    const float m_t = m_tab[t], var_t = var_tab[t];
    const float sig_obj = sqrtf(var_t);
    float m_nt = 0.f, sigma_t = 0.f, coef = 0.f;
    if (!is_last) {
        m_nt = m_tab[t_next];
        const float var_nt = var_tab[t_next];
        // sigma2_t = (var_t - var_nt * (1 - m_t)**2 / (1 - m_nt)**2) * var_nt / var_t       (BBM.py:194)
        const float a = (1.f - m_t) * (1.f - m_t);
        const float b = (1.f - m_nt) * (1.f - m_nt);
        const float sigma2 = (var_t - var_nt * a / b) * var_nt / var_t;
        sigma_t = sqrtf(sigma2) * eta;
        coef = sqrtf((var_nt - sigma2) / var_t);
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float xt = x_t[i], yy = y[i];
        float x0r = predict_x0_one(objective, xt, yy, pred[i], m_t, sig_obj);
        if (clip) x0r = fminf(fmaxf(x0r, -1.f), 1.f);
        x0_recon[i] = x0r;
        if (is_last) {
            x_next[i] = x0r;
            if (x_next_alias) x_next_alias[i] = x0r;
        } else {
            // (1 - m_nt) x0 + m_nt y + sqrt((var_nt - sigma2)/var_t) (x_t - (1 - m_t) x0 - m_t y) + sigma_t eps
            const float mean = (1.f - m_nt) * x0r + m_nt * yy + coef * (xt - (1.f - m_t) * x0r - m_t * yy);
            const float xn = mean + sigma_t * noise[i];
            x_next[i] = xn;
            if (x_next_alias) x_next_alias[i] = xn;        // second copy: the caller's input buffer of the NEXT step (see the header)
        }
    }
}

__global__ void __launch_bounds__(256) loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           unsigned long long* __restrict__ partial, size_t count, int loss_type) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s += loss_type == 0 ? (double)fabsf(d) : (double)d * (double)d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sa_add(partial, (red[0] + red[1]) + (red[2] + red[3]));      // exact limb cell: any block order, same bits
}

__global__ void loss_final_kernel(const unsigned long long* __restrict__ partial, float* __restrict__ out, double inv_count) {
    out[0] = (float)(sa_load(partial) * inv_count);
}

// d loss / d pred (same NCHW layout as pred):  l1: sign(pred - target) / count ; l2: 2 (pred - target) / count ;
// times the upstream scalar gradient gscale[0] (read on the device: no host sync).
__global__ void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                const float* __restrict__ gscale, float* __restrict__ dpred, size_t total,
                                float inv_count, int loss_type) {
    const float g = gscale[0] * inv_count;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        dpred[i] = loss_type == 0 ? (d > 0.f ? g : (d < 0.f ? -g : 0.f)) : 2.f * d * g;
    }
}

// NCHW (a [+ b]) -> NHWC with zero channel padding.  One thread per output pixel-channel; reads are coalesced
// along w for each source plane (C is tiny here: 3..16), writes are contiguous.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                    float* __restrict__ out, int ldo, int Cpad, int HW, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const size_t pix = i / Cpad;
        const size_t n = pix / HW, p = pix - n * HW;
        float v = 0.f;
        if (c < Ca) v = a[(n * Ca + c) * HW + p];
        else if (c < Ca + Cb) v = b[(n * Cb + (c - Ca)) * HW + p];
        out[pix * ldo + c] = v;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int C, int HW,
                                    size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW;
        const size_t nc = i / HW;
        const size_t n = nc / C, c = nc - n * C;
        out[i] = x[(n * HW + p) * ldx + c];
    }
}

inline unsigned ew_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

}  // namespace

extern "C" int bbdm_bb_q_sample_f32(const float* x0, const float* y, const float* noise, const int64_t* t,
                                    const float* m_t, const float* variance_t, float* x_t, float* target, int N,
                                    int per_sample, int objective, void* stream) {
    BBDM_REQUIRE(x0 && y && noise && t && m_t && variance_t && x_t && target, "q_sample: null pointer");
    BBDM_REQUIRE(N > 0 && per_sample > 0 && objective >= 0 && objective <= 2, "q_sample: bad args");
    const size_t total = (size_t)N * per_sample;
    hipLaunchKernelGGL(q_sample_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x0, y, noise, t, m_t,
                       variance_t, x_t, target, per_sample, total, objective);
    BBDM_CHECK_LAUNCH("q_sample");
    return BBDM_OK;
}

extern "C" int bbdm_bb_predict_x0_f32(const float* x_t, const float* y, const float* pred, const int64_t* t,
                                      const float* m_t, const float* variance_t, float* x0_recon, int N,
                                      int per_sample, int objective, void* stream) {
    BBDM_REQUIRE(x_t && y && pred && t && m_t && variance_t && x0_recon, "predict_x0: null pointer");
    BBDM_REQUIRE(N > 0 && per_sample > 0 && objective >= 0 && objective <= 2, "predict_x0: bad args");
    const size_t total = (size_t)N * per_sample;
    hipLaunchKernelGGL(predict_x0_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x_t, y, pred, t,
                       m_t, variance_t, x0_recon, per_sample, total, objective);
    BBDM_CHECK_LAUNCH("predict_x0");
    return BBDM_OK;
}

extern "C" int bbdm_bb_p_sample_step_f32(const float* x_t, const float* y, const float* pred, const float* noise,
                                         const float* m_t, const float* variance_t, int t, int t_next, int is_last,
                                         float eta, int clip, int objective, float* x_next, float* x0_recon,
                                         float* x_next_alias, int N, int per_sample, void* stream) {
    BBDM_REQUIRE(x_t && y && pred && m_t && variance_t && x_next && x0_recon, "p_sample_step: null pointer");
    BBDM_REQUIRE(is_last || noise, "p_sample_step: noise required unless is_last");
    BBDM_REQUIRE(N > 0 && per_sample > 0 && objective >= 0 && objective <= 2 && t >= 0 && (is_last || t_next >= 0),
                 "p_sample_step: bad args");
    const size_t total = (size_t)N * per_sample;
    hipLaunchKernelGGL(p_step_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x_t, y, pred, noise,
                       m_t, variance_t, t, t_next, is_last, eta, clip, objective, x_next, x0_recon, x_next_alias, total);
    BBDM_CHECK_LAUNCH("p_sample_step");
    return BBDM_OK;
}

extern "C" int bbdm_bb_loss_f32(const float* a, const float* b, double* partial, float* out, size_t count,
                                int loss_type, void* stream) {
    BBDM_REQUIRE(a && b && partial && out && count > 0 && (loss_type == 0 || loss_type == 1), "loss: bad args");
    size_t blocks = (count + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    unsigned long long* cell = reinterpret_cast<unsigned long long*>(partial);     // 4 x 8 bytes (stats_acc.h), zeroed by the caller
    hipLaunchKernelGGL(loss_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, cell,
                       count, loss_type);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, cell, out, 1.0 / (double)count);
    BBDM_CHECK_LAUNCH("loss");
    return BBDM_OK;
}

extern "C" int bbdm_bb_loss_bwd_f32(const float* pred, const float* target, const float* gscale, float* dpred,
                                    size_t count, int loss_type, void* stream) {
    BBDM_REQUIRE(pred && target && gscale && dpred && count > 0 && (loss_type == 0 || loss_type == 1),
                 "loss_bwd: bad args");
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(ew_blocks(count)), dim3(256), 0, (hipStream_t)stream, pred, target, gscale,
                       dpred, count, 1.0f / (float)count, loss_type);
    BBDM_CHECK_LAUNCH("loss_bwd");
    return BBDM_OK;
}

extern "C" int bbdm_nchw_to_nhwc_f32(const float* a, int Ca, const float* b, int Cb, float* out, int ldo, int Cpad,
                                     int N, int H, int W, void* stream) {
    BBDM_REQUIRE(a && out && Ca > 0 && Cb >= 0 && (Cb == 0 || b), "nchw_to_nhwc: bad args");
    BBDM_REQUIRE(Cpad >= Ca + Cb && ldo >= Cpad && N > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad shape");
    const size_t total = (size_t)N * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, Ca, b, Cb, out,
                       ldo, Cpad, H * W, total);
    BBDM_CHECK_LAUNCH("nchw_to_nhwc");
    return BBDM_OK;
}

extern "C" int bbdm_nhwc_to_nchw_f32(const float* x, int ldx, float* out, int N, int H, int W, int C, void* stream) {
    BBDM_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0 && ldx >= C, "nhwc_to_nchw: bad args");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, C,
                       H * W, total);
    BBDM_CHECK_LAUNCH("nhwc_to_nchw");
    return BBDM_OK;
}
