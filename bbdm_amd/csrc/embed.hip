// embed.hip -- timestep embedding and the M = batch dense layers of the conditioning path.
//
// Replaces timestep_embedding (util.py:151-171), time_embed = Linear -> SiLU -> Linear (openaimodel.py:511-516,735)
// and every ResBlock's emb_layers = SiLU -> Linear (openaimodel.py:221-227,267).  The 21 per-block projections
// share one input, so the host concatenates their weights once and issues ONE call ([N,512] x [512, sum 2*Cout]).
#include "common.h"

namespace {

__global__ void temb_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ emb,
                            int N, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * dim) return;
    const int n = i / dim, j = i - n * dim;
    float v = 0.f;
    if (j < 2 * half) {
        const int k = j < half ? j : j - half;
        // freqs is the host-computed table: the reference evaluates it with torch.exp on the CPU and then moves
        // it to the device (util.py:160-163), so a device expf would be one more rounding away from it.
        const float arg = (float)t[n] * freqs[k];
        v = j < half ? cosf(arg) : sinf(arg);
    }
    emb[i] = v;
}

// y[n][o] = act_out( b[o] + sum_i act_in(x[n][i]) w[o][i] ).  Block: 256 threads; x (all N rows) staged in LDS with
// act_in applied; lane <-> (row n, output o): NB = pow2 >= N rows share a wave, the wave's 64/NB outputs each stream
// their own weight row with float4 loads (every 128-B line is consumed in full over 8 iterations -> L1 hits).
template <int VEC>
__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int N, int In,
                                                     int Out, int NBl, int act_in, int act_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [N][In + 4]
    const int pitch = In + 4;
    for (int i = threadIdx.x; i < N * In; i += 256) {
        const int n = i / In, k = i - n * In;
        float v = x[i];
        if (act_in) v = silu_f(v);
        xs[n * pitch + k] = v;
    }
    __syncthreads();
    const int NB = 1 << NBl;
    const int n = threadIdx.x & (NB - 1);
    const int o = blockIdx.x * (256 >> NBl) + (threadIdx.x >> NBl);
    if (n >= N || o >= Out) return;
    const float* wr = w + (size_t)o * In;
    const float* xr = xs + n * pitch;
    float acc = 0.f;
    if (VEC) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < In; k += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + k);
            const float4 xv = *reinterpret_cast<const float4*>(xr + k);
            a0 = fmaf(wv.x, xv.x, a0);
            a1 = fmaf(wv.y, xv.y, a1);
            a2 = fmaf(wv.z, xv.z, a2);
            a3 = fmaf(wv.w, xv.w, a3);
        }
        acc = (a0 + a1) + (a2 + a3);
    } else {
        for (int k = 0; k < In; ++k) acc = fmaf(wr[k], xr[k], acc);
    }
    if (b) acc += b[o];
    if (act_out) acc = silu_f(acc);
    y[(size_t)n * Out + o] = acc;
}

// ---- backward of linear_kernel (training path; M = batch, so these are tiny) ---------------------------------------
// dx_pre[n][i] = act_in'(x[n][i]) * sum_o dy[n][o] w[o][i].  grid = (In/256, o-chunks); each thread owns one input
// column i for ALL rows n (<= 64 accumulators), so every weight element is read exactly once; the o-chunks write
// partials that a second kernel sums in a fixed order (deterministic).
constexpr int LB_OCH = 256;      // outputs per chunk

__global__ void __launch_bounds__(256) linear_bwd_x_partial_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                   float* __restrict__ part, int N, int In, int Out) {
    __shared__ __attribute__((aligned(16))) float dys[64 * LB_OCH];           // [n][o in chunk]
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * LB_OCH;
    const int on = min(LB_OCH, Out - o0);
    for (int k = threadIdx.x; k < N * LB_OCH; k += 256) {
        const int n = k / LB_OCH, o = k - n * LB_OCH;
        dys[k] = o < on ? dy[(size_t)n * Out + o0 + o] : 0.f;
    }
    __syncthreads();
    if (i >= In) return;
    float acc[64];
#pragma unroll
    for (int n = 0; n < 64; ++n) acc[n] = 0.f;
    // four outputs per step: one broadcast ds_read_b128 of dy per row feeds four FMAs (one LDS read per FMA made this kernel
    // LDS-issue-bound: 490 us for the 51 MB FiLM projection of the LBBDM-f4 UNet); same accumulation order as the scalar loop
    int o = 0;
    for (; o + 4 <= on; o += 4) {
        const float w0 = w[(size_t)(o0 + o) * In + i], w1 = w[(size_t)(o0 + o + 1) * In + i];
        const float w2 = w[(size_t)(o0 + o + 2) * In + i], w3 = w[(size_t)(o0 + o + 3) * In + i];
#pragma unroll
        for (int n = 0; n < 64; ++n)
            if (n < N) {
                const float4 d = *reinterpret_cast<const float4*>(&dys[n * LB_OCH + o]);
                acc[n] = fmaf(d.w, w3, fmaf(d.z, w2, fmaf(d.y, w1, fmaf(d.x, w0, acc[n]))));
            }
    }
    for (; o < on; ++o) {
        const float wv = w[(size_t)(o0 + o) * In + i];
#pragma unroll
        for (int n = 0; n < 64; ++n)
            if (n < N) acc[n] = fmaf(dys[n * LB_OCH + o], wv, acc[n]);
    }
#pragma unroll
    for (int n = 0; n < 64; ++n)
        if (n < N) part[((size_t)blockIdx.y * N + n) * In + i] = acc[n];
}

__global__ void linear_bwd_x_final_kernel(const float* __restrict__ part, const float* __restrict__ x_pre,
                                          float* __restrict__ dx, int chunks, int N, int In, int act_in) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * In) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)c * N * In + idx];
    if (act_in) {
        const float v = x_pre[idx];
        const float sg = 1.0f / (1.0f + expf(-v));
        s *= sg * (1.0f + v * (1.0f - sg));
    }
    dx[idx] = s;
}

// dw[o][i] = sum_n dy[n][o] act_in(x[n][i]),  db[o] = sum_n dy[n][o].  Block = 8 outputs x all inputs; x staged in LDS.
__global__ void __launch_bounds__(256) linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ db, int N, int In,
                                                           int Out, int act_in) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [N][In]
    for (int k = threadIdx.x; k < N * In; k += 256) {
        float v = x[k];
        if (act_in) v = silu_f(v);
        xs[k] = v;
    }
    __syncthreads();
    const int o0 = blockIdx.x * 8;
    for (int oo = 0; oo < 8; ++oo) {
        const int o = o0 + oo;
        if (o >= Out) break;
        for (int i = threadIdx.x; i < In; i += 256) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s = fmaf(dy[(size_t)n * Out + o], xs[n * In + i], s);
            dw[(size_t)o * In + i] = s;
        }
        if (db && threadIdx.x == 0) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += dy[(size_t)n * Out + o];
            db[o] = s;
        }
    }
}

}  // namespace

extern "C" size_t bbdm_linear_bwd_workspace_floats(int N, int In, int Out) {
    return (size_t)cdiv(Out, LB_OCH) * N * In;
}

extern "C" int bbdm_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                   float* ws, int N, int In, int Out, int act_in, void* stream) {
    BBDM_REQUIRE(dy && x && w && dw && ws && N > 0 && N <= 64 && In > 0 && Out > 0, "linear_bwd: bad args (N <= 64)");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)N * In * sizeof(float);
    BBDM_REQUIRE(lds <= 160 * 1024, "linear_bwd: N*In too large for LDS staging");
    static size_t lds_set_dev[BBDM_MAX_DEVICES] = {};
    size_t& lds_set = lds_set_dev[bbdm_device_slot()];
    if (lds > 64 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bwd_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            bbdm_set_error("linear_bwd: hipFuncSetAttribute failed");
            return BBDM_E_LAUNCH;
        }
        lds_set = lds;
    }
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(cdiv(Out, 8)), dim3(256), lds, st, dy, x, dw, db, N, In, Out, act_in);
    if (dx) {
        const int chunks = cdiv(Out, LB_OCH);
        hipLaunchKernelGGL(linear_bwd_x_partial_kernel, dim3(cdiv(In, 256), chunks), dim3(256), 0, st, dy, w, ws, N, In, Out);
        hipLaunchKernelGGL(linear_bwd_x_final_kernel, dim3(cdiv(N * In, 256)), dim3(256), 0, st, ws, x, dx, chunks, N, In,
                           act_in);
    }
    BBDM_CHECK_LAUNCH("linear_bwd");
    return BBDM_OK;
}

extern "C" int bbdm_timestep_embedding_f32(const int64_t* t, const float* freqs, float* emb, int N, int dim,
                                           void* stream) {
    BBDM_REQUIRE(t && freqs && emb && N > 0 && dim > 0, "timestep_embedding: bad args");
    const int total = N * dim;
    hipLaunchKernelGGL(temb_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, t, freqs, emb, N, dim);
    BBDM_CHECK_LAUNCH("timestep_embedding");
    return BBDM_OK;
}

extern "C" int bbdm_linear_f32(const float* x, const float* w, const float* b, float* y, int N, int In, int Out,
                               int act_in, int act_out, void* stream) {
    BBDM_REQUIRE(x && w && y && N > 0 && In > 0 && Out > 0, "linear: bad args");
    BBDM_REQUIRE(N <= 64, "linear: N=%d rows > 64 (chunk the batch on the host)", N);
    const size_t lds = (size_t)N * (In + 4) * sizeof(float);
    BBDM_REQUIRE(lds <= 160 * 1024, "linear: N*In too large for LDS staging (%zu B)", lds);
    const int NBl = ilog2(ceil_pow2(N));
    const int outs_per_block = 256 >> NBl;
    const bool vec = (In % 4 == 0) && (((uintptr_t)w & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    static size_t lds_set_dev[BBDM_MAX_DEVICES][2] = {};
    size_t (&lds_set)[2] = lds_set_dev[bbdm_device_slot()];
    if (lds > 64 * 1024 && lds > lds_set[vec]) {
        const void* f = vec ? reinterpret_cast<const void*>(linear_kernel<1>) : reinterpret_cast<const void*>(linear_kernel<0>);
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            bbdm_set_error("linear: hipFuncSetAttribute(%zu) failed", lds);
            return BBDM_E_LAUNCH;
        }
        lds_set[vec] = lds;
    }
    const dim3 grid(cdiv(Out, outs_per_block));
    if (vec)
        hipLaunchKernelGGL(linear_kernel<1>, grid, dim3(256), lds, st, x, w, b, y, N, In, Out, NBl, act_in, act_out);
    else
        hipLaunchKernelGGL(linear_kernel<0>, grid, dim3(256), lds, st, x, w, b, y, N, In, Out, NBl, act_in, act_out);
    BBDM_CHECK_LAUNCH("linear");
    return BBDM_OK;
}
