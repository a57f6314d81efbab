// firststage.hip -- the pieces of the VQGAN first stage that the UNet kernels do not already provide.
//
// model/VQGAN/model.py:140-192 (AttnBlock): ONE head over all C (128..512) channels, softmax over the T = H*W pixels.  The
// streaming-softmax kernel (attention.hip) keeps a query's whole channel vector in registers and stops at 64 channels per
// head; here the scores are materialised like the reference does (w_ = bmm(q, k), [N, T, T]) by the batched GEMM of
// conv_igemm.hip -- the K (then V^T) rows of each image packed as that image's "weights" -- with this file's row softmax
// in between.  The first stage runs once per batch (2 encodes + 1 decode per 200 UNet calls).
//
// model/VQGAN/quantize.py:271-286 (VectorQuantizer2.forward): nearest codebook entry,
//     d = sum(z^2, 1) + sum(e^2, 1) - 2 z e^T;  argmin_j d
// e_dim is 3..8, so one thread evaluates a pixel against the whole codebook held in LDS.
#include "common.h"

namespace {

// in-place softmax(scale * s) over each row of [rows][T] (pitch ld); one wavefront per row, three passes over L2-resident
// data (a 4096-float row is 16 KB)
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ s, int ld, long long rows, int T, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = s + (size_t)row * ld;
    float m = -INFINITY;
    for (int j = lane; j < T; j += 64) m = fmaxf(m, r[j] * scale);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
        const float e = expf(r[j] * scale - m);
        r[j] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
    for (int j = lane; j < T; j += 64) r[j] *= inv;
}

// codebook [n_e][D] in LDS (n_e * D floats); thread = one latent pixel.  The arithmetic follows the reference expression
// term by term in fp32: zz = sum_k z_k^2, ee_j = sum_k e_jk^2 (both sequential), dot = fma chain over k starting from the
// first product (what an fp32 GEMM micro-kernel computes for K <= 8), d = (zz + ee_j) - 2 * dot; first minimum wins.
template <int D>
__global__ void __launch_bounds__(256) vq_argmin_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ cb,
                                                        long long* __restrict__ idx, float* __restrict__ zq, int ldq,
                                                        long long pixels, int n_e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* e = smem;                   // [n_e][D]
    float* ee = smem + (size_t)n_e * D;    // [n_e]
    for (int i = threadIdx.x; i < n_e * D; i += 256) e[i] = cb[i];
    __syncthreads();
    for (int j = threadIdx.x; j < n_e; j += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) s += e[j * D + k] * e[j * D + k];
        ee[j] = s;
    }
    __syncthreads();
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
        float zv[D];
        float zz = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            zv[k] = z[(size_t)p * ldz + k];
            zz += zv[k] * zv[k];
        }
        float best = INFINITY;
        int bj = 0;
        for (int j = 0; j < n_e; ++j) {
            float dot = zv[0] * e[j * D];
#pragma unroll
            for (int k = 1; k < D; ++k) dot = fmaf(zv[k], e[j * D + k], dot);
            const float d = (zz + ee[j]) - 2.f * dot;
            if (d < best) { best = d; bj = j; }
        }
        if (idx) idx[p] = bj;
        if (zq) {
#pragma unroll
            for (int k = 0; k < D; ++k) zq[(size_t)p * ldq + k] = e[bj * D + k];
        }
    }
}

}  // namespace

extern "C" int bbdm_softmax_rows_f32(float* s, int ld, long long rows, int T, float scale, void* stream) {
    BBDM_REQUIRE(s && rows > 0 && T > 0 && ld >= T, "softmax_rows: bad args");
    BBDM_REQUIRE((rows + 3) / 4 < (1ll << 31), "softmax_rows: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, ld, rows, T,
                       scale);
    BBDM_CHECK_LAUNCH("softmax_rows");
    return BBDM_OK;
}

extern "C" int bbdm_vq_nearest_f32(const float* z, int ldz, const float* codebook, long long* indices, float* zq, int ldq,
                                   long long pixels, int n_e, int e_dim, void* stream) {
    BBDM_REQUIRE(z && codebook && (indices || zq) && pixels > 0 && n_e > 0, "vq_nearest: bad args");
    BBDM_REQUIRE(e_dim >= 1 && e_dim <= 8 && ldz >= e_dim && (!zq || ldq >= e_dim), "vq_nearest: e_dim=%d (1..8)", e_dim);
    const size_t lds = ((size_t)n_e * e_dim + n_e) * sizeof(float);
    BBDM_REQUIRE(lds <= 160 * 1024, "vq_nearest: codebook of %d x %d floats does not fit the LDS", n_e, e_dim);
    long long blocks = (pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    static size_t lds_set_dev[BBDM_MAX_DEVICES][9] = {};
#define BBDM_VQ(D)                                                                                                          \
    case D: {                                                                                                               \
        size_t& set = lds_set_dev[bbdm_device_slot()][D];                                                                   \
        if (lds > 64 * 1024 && lds > set) {                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<D>),                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {                  \
                bbdm_set_error("vq_nearest: hipFuncSetAttribute(%zu B LDS) failed", lds);                                   \
                return BBDM_E_LAUNCH;                                                                                       \
            }                                                                                                               \
            set = lds;                                                                                                      \
        }                                                                                                                   \
        hipLaunchKernelGGL(vq_argmin_kernel<D>, dim3((unsigned)blocks), dim3(256), lds, st, z, ldz, codebook, indices, zq,   \
                           ldq, pixels, n_e);                                                                               \
        break;                                                                                                              \
    }
    switch (e_dim) {
        BBDM_VQ(1) BBDM_VQ(2) BBDM_VQ(3) BBDM_VQ(4) BBDM_VQ(5) BBDM_VQ(6) BBDM_VQ(7) BBDM_VQ(8)
    }
#undef BBDM_VQ
    BBDM_CHECK_LAUNCH("vq_nearest");
    return BBDM_OK;
}
