// egress.hip -- sample egress: a batch of generated images, fp32 NCHW in [-1, 1] on the device, to
// uint8 HWC pixels in ONE pass, ready for a single device-to-host copy.
//
// Replaces, per image, the six ATen kernels + the blocking .to('cpu') of runners/utils.py:67-74 (save_single_image, called
// 1-3x per image from BBDMRunner.sample_to_eval, BBDMRunner.py:242-253):
//     image.mul_(0.5).add_(0.5).clamp_(0, 1.)                       (to_normal)
//     image.mul_(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8)
// The arithmetic is that sequence operation by operation in fp32 (this file is built with -ffp-contract=off), the final
// conversion truncates like torch's float -> uint8 cast: the bytes are IDENTICAL to the reference's (tests compare them
// bit for bit, and the PNG files byte for byte).  HBM-bound: reads 4 B, writes 1 B per element.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned char to_u8(float v, int to_normal) {
    if (to_normal) {
        v = v * 0.5f;
        v = v + 0.5f;
        v = fminf(fmaxf(v, 0.f), 1.f);
    }
    v = v * 255.f;
    v = v + 0.5f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (unsigned char)(int)v;               // v in [0, 255]: truncation toward zero, as at::native's cast
}

// one thread = one output pixel (all C channels): reads are coalesced along w in each channel plane, the C bytes of a
// pixel are written together.  C is 1..4 for images; larger C takes the generic loop.
__global__ void __launch_bounds__(256) images_to_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                           int C, int HW, size_t npix, int to_normal) {
    for (size_t p = blockIdx.x * (size_t)256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        const size_t n = p / HW, hw = p - n * HW;
        const float* src = x + n * (size_t)C * HW + hw;
        unsigned char* dst = out + p * C;
        if (C == 3) {
            const unsigned char r = to_u8(src[0], to_normal), g = to_u8(src[(size_t)HW], to_normal),
                                b = to_u8(src[2 * (size_t)HW], to_normal);
            dst[0] = r; dst[1] = g; dst[2] = b;
        } else {
            for (int c = 0; c < C; ++c) dst[c] = to_u8(src[(size_t)c * HW], to_normal);
        }
    }
}

}  // namespace

extern "C" int bbdm_images_to_u8_f32(const float* x_nchw, unsigned char* out_nhwc, int N, int C, int H, int W, int to_normal,
                                     void* stream) {
    BBDM_REQUIRE(x_nchw && out_nhwc && N > 0 && C > 0 && H > 0 && W > 0, "images_to_u8: bad args");
    const size_t npix = (size_t)N * H * W;
    size_t blocks = (npix + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(images_to_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_nchw, out_nhwc, C,
                       H * W, npix, to_normal);
    BBDM_CHECK_LAUNCH("images_to_u8");
    return BBDM_OK;
}
