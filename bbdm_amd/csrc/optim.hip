// optim.hip -- the optimizer side of the training step as ONE HBM-bound pass.
//
// Replaces, per optimizer step of the 237 M-parameter UNet (248 tensors):
//   * torch.optim.Adam.step()  (runners/utils.py:48-51: Adam(lr, weight_decay, betas=(beta1, 0.999)), eps 1e-8, no amsgrad)
//     -- ~10 ATen foreach kernels over p, g, m, v in the reference's torch 1.12 (``_single_tensor_adam``: 248 x ~8 launches);
//   * EMA.update()             (runners/base/EMA.py:21-29: shadow = (1 - d) * p + d * shadow, or shadow = p before
//     start_ema_step) -- 248 x 4 launches + a clone each;
// with one launch over a table of (param, grad, exp_avg, exp_avg_sq, shadow) chunks: p, g, m, v are read once and p, m, v
// (and the EMA shadow, when due) written once: 7 (+2) x 4 B per parameter = 6.6 (8.5) GB per step at 237 M parameters.
// The tensors stay where torch put them (leaf nn.Parameters, .grad views of the backward plan's flat buffer or DDP bucket
// views): the table holds raw pointers, so no flattening / re-pointing of parameters is needed.
//
// Arithmetic follows the single-tensor path of torch.optim.Adam in torch >= 2.0 operation by operation (this file is built with
// -ffp-contract=off, like bridge.hip) -- the torch the tests compare with.  The reference's torch 1.12 updates the first moment as
// exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1): the same value up to one fp32 rounding per step, not bit for bit.
//     g' = g + wd * p                                  (weight_decay != 0)
//     m  = lerp(m, g', 1 - b1)                         (exp_avg.lerp_(grad, 1 - beta1); at::lerp's two-branch formula)
//     v  = v * b2 + (1 - b2) * g' * g'                 (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2))
//     p  = p - (lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps)),   bc1 = 1 - b1^step, bc2 = 1 - b2^step  (host doubles -> fp32)
#include "common.h"

namespace {

constexpr int CHUNK = 16384;       // elements per table entry = per workgroup (64 per thread)

struct OptArgs {
    const BbdmOptChunk* table;
    float lr_over_bc1, sqrt_bc2, b1c, b2, b2c, eps, wd;     // b1c = 1 - beta1, b2c = 1 - beta2
    float ema_decay, ema_c;                                       // ema_c = 1 - decay
    int do_adam, ema_mode;                                        // ema_mode: 0 none, 1 decay, 2 copy
};

__device__ __forceinline__ void adam_one(const OptArgs& a, float& p, float g, float& m, float& v) {
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);          // grad.add(param, alpha=wd): one rounding in ATen's kernels
    // at::lerp: |w| < 0.5 ? a + w (b - a) : b - (b - a) (1 - w),  w = 1 - beta1
    m = a.b1c < 0.5f ? m + a.b1c * (g - m) : g - (g - m) * (1.f - a.b1c);
    v = v * a.b2 + a.b2c * g * g;
    const float denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
    p = p - a.lr_over_bc1 * (m / denom);
}

__global__ void __launch_bounds__(256) adam_ema_kernel(const OptArgs a) {
    const BbdmOptChunk c = a.table[blockIdx.x];
    float* __restrict__ p = c.param;
    const float* __restrict__ g = c.grad;
    float* __restrict__ m = c.exp_avg;
    float* __restrict__ v = c.exp_avg_sq;
    float* __restrict__ s = c.shadow;
    const int n = c.n;
    const bool adam = a.do_adam && g != nullptr;      // a parameter without a gradient is skipped, as torch does
    const bool ema = a.ema_mode != 0 && s != nullptr;
    const uintptr_t al = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)s;
    if ((al & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pv = reinterpret_cast<const float4*>(p)[i];
            if (adam) {
                const float4 gv = reinterpret_cast<const float4*>(g)[i];
                float4 mv = reinterpret_cast<const float4*>(m)[i];
                float4 vv = reinterpret_cast<const float4*>(v)[i];
                adam_one(a, pv.x, gv.x, mv.x, vv.x);
                adam_one(a, pv.y, gv.y, mv.y, vv.y);
                adam_one(a, pv.z, gv.z, mv.z, vv.z);
                adam_one(a, pv.w, gv.w, mv.w, vv.w);
                reinterpret_cast<float4*>(m)[i] = mv;
                reinterpret_cast<float4*>(v)[i] = vv;
                reinterpret_cast<float4*>(p)[i] = pv;
            }
            if (ema) {
                float4 sv = pv;
                if (a.ema_mode == 1) {
                    const float4 o = reinterpret_cast<const float4*>(s)[i];
                    sv.x = a.ema_c * pv.x + a.ema_decay * o.x;
                    sv.y = a.ema_c * pv.y + a.ema_decay * o.y;
                    sv.z = a.ema_c * pv.z + a.ema_decay * o.z;
                    sv.w = a.ema_c * pv.w + a.ema_decay * o.w;
                }
                reinterpret_cast<float4*>(s)[i] = sv;
            }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            float pv = p[i];
            if (adam) {
                float mv = m[i], vv = v[i];
                adam_one(a, pv, g[i], mv, vv);
                m[i] = mv; v[i] = vv; p[i] = pv;
            }
            if (ema) s[i] = a.ema_mode == 1 ? a.ema_c * pv + a.ema_decay * s[i] : pv;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            float pv = p[i];
            if (adam) {
                float mv = m[i], vv = v[i];
                adam_one(a, pv, g[i], mv, vv);
                m[i] = mv; v[i] = vv; p[i] = pv;
            }
            if (ema) s[i] = a.ema_mode == 1 ? a.ema_c * pv + a.ema_decay * s[i] : pv;
        }
    }
}

}  // namespace

extern "C" int bbdm_opt_chunk_elems(void) { return CHUNK; }

extern "C" int bbdm_adam_ema_step_f32(const BbdmOptChunk* table, int nchunks, int do_adam, double lr, double beta1,
                                      double beta2, double eps, double weight_decay, long long step, int ema_mode,
                                      double ema_decay, void* stream) {
    BBDM_REQUIRE(table && nchunks > 0, "adam_ema: empty chunk table");
    BBDM_REQUIRE(do_adam || ema_mode, "adam_ema: nothing to do");
    BBDM_REQUIRE(ema_mode >= 0 && ema_mode <= 2, "adam_ema: ema_mode=%d (0 none, 1 decay, 2 copy)", ema_mode);
    BBDM_REQUIRE(!do_adam || (step >= 1 && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.),
                 "adam_ema: bad hyper-parameters (step=%lld beta1=%g beta2=%g eps=%g)", step, beta1, beta2, eps);
    OptArgs a;
    a.table = table;
    // hyper-parameters arrive as the Python doubles torch works with; every derived scalar is formed in double and
    // rounded to fp32 once, which is what a Python-number operand of an fp32 tensor op undergoes
    const double bc1 = do_adam ? 1.0 - pow(beta1, (double)step) : 1.0;
    const double bc2 = do_adam ? 1.0 - pow(beta2, (double)step) : 1.0;
    a.lr_over_bc1 = (float)(lr / bc1);
    a.sqrt_bc2 = (float)sqrt(bc2);
    a.b1c = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.b2c = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.wd = (float)weight_decay;
    a.ema_decay = (float)ema_decay;
    a.ema_c = (float)(1.0 - ema_decay);
    a.do_adam = do_adam;
    a.ema_mode = ema_mode;
    hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, a);
    BBDM_CHECK_LAUNCH("adam_ema");
    return BBDM_OK;
}
