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
// e_dim is 3..8: one thread evaluates a pixel against the codebook streamed through LDS.
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

// thread = one latent pixel; the codebook streams through LDS in chunks of VQ_CHUNK rows (the f8 / f16 codebooks, 16384 x 4..8
// floats, do not fit it whole).  The arithmetic follows the reference expression term by term in fp32: zz = sum_k z_k^2,
// ee_j = sum_k e_jk^2 (both sequential), dot = fma chain over k starting from the first product (what an fp32 GEMM
// micro-kernel computes for K <= 8), d = (zz + ee_j) - 2 * dot; the first minimum wins (chunks are visited in index order).
constexpr int VQ_CHUNK = 1024;

template <int D>
__global__ void __launch_bounds__(256) vq_argmin_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ cb,
                                                        long long* __restrict__ idx, float* __restrict__ zq, int ldq,
                                                        long long pixels, int n_e) {
    __shared__ float e[VQ_CHUNK * D];
    __shared__ float ee[VQ_CHUNK];
    for (long long base = blockIdx.x * 256ll; base < pixels; base += (long long)gridDim.x * 256) {   // uniform per block
        const long long p = base + threadIdx.x;
        const bool valid = p < pixels;
        float zv[D];
        float zz = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            zv[k] = valid ? z[(size_t)p * ldz + k] : 0.f;
            zz += zv[k] * zv[k];
        }
        float best = INFINITY;
        int bj = 0;
        for (int c0 = 0; c0 < n_e; c0 += VQ_CHUNK) {
            const int cn = min(VQ_CHUNK, n_e - c0);
            __syncthreads();                               // the previous chunk is no longer being read
            for (int i = threadIdx.x; i < cn * D; i += 256) e[i] = cb[(size_t)c0 * D + i];
            __syncthreads();
            for (int j = threadIdx.x; j < cn; j += 256) {
                float s2 = 0.f;
#pragma unroll
                for (int k = 0; k < D; ++k) s2 += e[j * D + k] * e[j * D + k];
                ee[j] = s2;
            }
            __syncthreads();
            for (int j = 0; j < cn; ++j) {
                float dot = zv[0] * e[j * D];
#pragma unroll
                for (int k = 1; k < D; ++k) dot = fmaf(zv[k], e[j * D + k], dot);
                const float d = (zz + ee[j]) - 2.f * dot;
                if (d < best) { best = d; bj = c0 + j; }
            }
        }
        if (valid) {
            if (idx) idx[p] = bj;
            if (zq) {
#pragma unroll
                for (int k = 0; k < D; ++k) zq[(size_t)p * ldq + k] = cb[(size_t)bj * D + k];
            }
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
    long long blocks = (pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
#define BBDM_VQ(D)                                                                                                          \
    case D:                                                                                                                 \
        hipLaunchKernelGGL(vq_argmin_kernel<D>, dim3((unsigned)blocks), dim3(256), 0, st, z, ldz, codebook, indices, zq, ldq, \
                           pixels, n_e);                                                                                    \
        break;
    switch (e_dim) {
        BBDM_VQ(1) BBDM_VQ(2) BBDM_VQ(3) BBDM_VQ(4) BBDM_VQ(5) BBDM_VQ(6) BBDM_VQ(7) BBDM_VQ(8)
    }
#undef BBDM_VQ
    BBDM_CHECK_LAUNCH("vq_nearest");
    return BBDM_OK;
}
