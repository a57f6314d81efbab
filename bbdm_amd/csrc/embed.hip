// embed.hip -- timestep embedding and the M = batch dense layers of the conditioning path.
//
// Replaces timestep_embedding (util.py:151-171), time_embed = Linear -> SiLU -> Linear (openaimodel.py:511-516,735)
// and every ResBlock's emb_layers = SiLU -> Linear (openaimodel.py:221-227,267).  The 21 per-block projections
// share one input, so the host concatenates their weights once and issues ONE call ([N,512] x [512, sum 2*Cout]).
#include "common.h"
#include <stdlib.h>

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

// act_in(x) [rows x In] -> LDS (pitch floats per row; rows >= N zero), by the 256 threads of the workgroup.  Every workgroup of a
// launch stages the same x: 16-byte loads where In allows (the scalar loop was 64 dependent-free loads + 64 SiLUs per thread and a
// third of the FiLM projection's 47 us at batch 32).
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, float* xs, int rows, int N, int In, int pitch, int act_in, int tid) {
    if ((In & 3) == 0 && (((uintptr_t)x & 15) == 0)) {
        const int In4 = In >> 2;
        for (int k = tid; k < rows * In4; k += 256) {
            const int n = k / In4, i = (k - n * In4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < N) {
                v = *reinterpret_cast<const float4*>(x + (size_t)n * In + i);
                if (act_in) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            }
            float* d = xs + n * pitch + i;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        return;
    }
    for (int k = tid; k < rows * In; k += 256) {
        const int n = k / In, i = k - n * In;
        float v = 0.f;
        if (n < N) {
            v = x[(size_t)n * In + i];
            if (act_in) v = silu_f(v);
        }
        xs[n * pitch + i] = v;
    }
}

// The same product on the f32 matrix core (In % 8 == 0): the rows of w are the B operand straight from global memory -- lane (o, half)
// streams ITS row, 16 bytes per four MFMA steps (the k pair of a step is (8 q + e, 8 q + 4 + e): both halves read contiguous float4s) --
// and act_in(x) is the A operand from LDS (pitch In + 1).  A wave owns 32 outputs for all rows; the 51 MB FiLM projection of the
// LBBDM-f4 / pixel UNets moves once at HBM rate (linear_kernel: one load instruction per 2 weight rows, 0.14 ms at batch 32).
template <int MB>
__global__ void __launch_bounds__(256) linear_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y, int N, int In,
                                                          int Out, int act_in, int act_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [MB * 32][In + 1], rows >= N zero
    const int pitch = In + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    stage_rows(x, xs, MB * 32, N, In, pitch, act_in, tid);
    __syncthreads();
    const int otiles = (Out + 31) / 32, nq = In / 8;
    const float* a0p = xs + l31 * pitch + 4 * hi;
    for (int ot = blockIdx.x * 4 + wave; ot < otiles; ot += gridDim.x * 4) {
        const int o = ot * 32 + l31;
        const bool ov = o < Out;
        const float* wr = w + (size_t)(ov ? o : Out - 1) * In + 4 * hi;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        // eight float4 (64 k) of the row per group, the next group requested before this one is multiplied: 256 B per lane in flight
        // (with four the loop was one HBM round trip per 16 MFMAs: 52 us for the 51 MB projection)
        float4 wv[2][8];
        auto request = [&](int q, float4 (&dst)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int qq = q + u < nq ? q + u : nq - 1;                    // clamped address + select: no branch between the loads
                const float4 v = *reinterpret_cast<const float4*>(wr + qq * 8);
                dst[u] = (ov && q + u < nq) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto multiply = [&](int q, const float4 (&src)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (q + u < nq) {
                    const float* ap = a0p + (q + u) * 8;
                    const float bs[4] = {src[u].x, src[u].y, src[u].z, src[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[e], bs[e], acc0, 0, 0, 0);
                        if (MB == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[32 * pitch + e], bs[e], acc1, 0, 0, 0);
                    }
                }
            }
        };
        request(0, wv[0]);
        for (int q = 0; q < nq; q += 16) {
            if (q + 8 < nq) request(q + 8, wv[1]);
            multiply(q, wv[0]);
            if (q + 16 < nq) request(q + 16, wv[0]);
            if (q + 8 < nq) multiply(q + 8, wv[1]);
        }
        if (ov) {
            const float bo = b ? b[o] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n < N) {
                    float v = acc0[r] + bo;
                    if (act_out) v = silu_f(v);
                    y[(size_t)n * Out + o] = v;
                }
                if (MB == 2 && 32 + n < N) {
                    float v = acc1[r] + bo;
                    if (act_out) v = silu_f(v);
                    y[(size_t)(32 + n) * Out + o] = v;
                }
            }
        }
    }
}

// ---- the same product on PACKED weights (inference plans: the weights are packed once, round 4) ---------------------------------------
// linear_mfma_kernel lets every lane stream its own weight row: 25 000 concurrent 2-KB rows, 32 B at a time each -- the 51 MB FiLM
// projection moves at 1.0 TB/s (51 us per forward of every configuration; 36 us average over the three linears of an LBBDM-f16 step).
// Here the weights are packed into 4 KB blocks [32 outputs x 32 k] that are already the LDS image the B fragments are read from
// ([q = k / 8][half][output] x 16 B: the 64 lanes of a fragment read sit on consecutive 16-B slots), one block after the other per
// output tile: a wave streams ITS tile's blocks as ONE sequential stream of 1 KB LDS-DMA copies, four blocks deep, into LDS it shares
// with nobody (no barrier in the loop: counted vmcnt only).  Same MFMA steps in the same order as linear_mfma_kernel: the same bits.
__device__ __forceinline__ unsigned lds_address(void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)p; }
// four 16-byte LDS reads 1 KB apart + the wait for them, as ONE asm statement (tools/hipemu replaces it with four plain loads)
#ifndef BBDM_LDS_READ4_1K
#define BBDM_LDS_READ4_1K(d0, d1, d2, d3, addr)                                                                                   \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"                 \
                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"                                                        \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(addr) : "memory")
#endif
constexpr int LP_KB = 32;                      // k per block
constexpr int LP_BLOCK = 32 * LP_KB * 4;       // bytes per block
constexpr int LP_NS = 4;                       // blocks in the ring of one wave

__global__ void linear_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int Out, int In, int otiles) {
    const int nkb = In / LP_KB;
    const size_t total = (size_t)otiles * nkb * (LP_BLOCK / 16);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % (LP_BLOCK / 16));                  // 16-B slot inside the block: ((q * 2 + half) * 32 + output)
        const size_t blk = i / (LP_BLOCK / 16);
        const int kb = (int)(blk % nkb), ot = (int)(blk / nkb);
        const int o = ot * 32 + (slot & 31), half = (slot >> 5) & 1, q = slot >> 6;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < Out) v = *reinterpret_cast<const float4*>(w + (size_t)o * In + kb * LP_KB + q * 8 + half * 4);
        *reinterpret_cast<float4*>(p + i * 4) = v;
    }
}

__global__ void __launch_bounds__(256) linear_packed_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wp,
                                                            const float* __restrict__ b, float* __restrict__ y, int N, int In, int Out,
                                                            int act_in, int act_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [32][In + 1] (rows >= N zero), then [4 waves][LP_NS][LP_BLOCK]
    const int pitch = In + 1;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int otiles = (Out + 31) / 32, nkb = In / LP_KB;
    const int ot = (int)blockIdx.x * 4 + wave;
    unsigned char* const ring = reinterpret_cast<unsigned char*>(xs) + (((size_t)32 * pitch * 4 + 15) & ~(size_t)15) + (size_t)wave * (LP_NS * LP_BLOCK);
    const unsigned char* const src = wp + (size_t)(ot < otiles ? ot : otiles - 1) * nkb * LP_BLOCK + lane * 16;
    auto issue = [&](int kb, unsigned char* st) {
#pragma unroll
        for (int c = 0; c < LP_BLOCK / 1024; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)kb * LP_BLOCK + c * 1024),
                                             (__attribute__((address_space(3))) void*)(st + c * 1024), 16, 0, 0);
    };
    // the first blocks are requested before x is staged: their latency hides under the staging
#pragma unroll
    for (int k = 0; k < LP_NS - 1; ++k)
        if (k < nkb) issue(k, ring + k * LP_BLOCK);
    stage_rows(x, xs, 32, N, In, pitch, act_in, tid);
    __syncthreads();
    if (ot >= otiles) {                                   // (a padding wave of the last workgroup: its copies must land before it ends)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* a0p = xs + l31 * pitch + 4 * hi;
    const unsigned ring0 = lds_address(ring) + lane * 16;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (int kb = 0; kb < nkb; ++kb) {
        // block kb has landed (the younger ones that were requested may stay in flight)
        const int younger = min(LP_NS - 2, nkb - 1 - kb);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned st = ring0 + (unsigned)((kb % LP_NS) * LP_BLOCK);
        f32x4 wv[LP_KB / 8];
        // (hand-written: a plain LDS load after an LDS-DMA copy is guarded with vmcnt(0).  The four reads AND their wait are ONE asm
        // statement with early-clobber outputs: the compiler cannot place a copy or a spill of a destination register between a read and
        // the wait it does not know about -- round-4 advisor finding)
        static_assert(LP_KB / 8 == 4, "the fragment reads below are written out for four 16-byte pieces");
        BBDM_LDS_READ4_1K(wv[0], wv[1], wv[2], wv[3], st);
        if (kb + LP_NS - 1 < nkb) issue(kb + LP_NS - 1, ring + ((kb + LP_NS - 1) % LP_NS) * LP_BLOCK);   // the stage block kb - 1 was read from
#pragma unroll
        for (int q = 0; q < LP_KB / 8; ++q) {
            const float* ap = a0p + kb * LP_KB + q * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[e], wv[q][e], acc, 0, 0, 0);
        }
    }
    const int o = ot * 32 + l31;
    if (o < Out) {
        const float bo = b ? b[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (n < N) {
                float v = acc[r] + bo;
                if (act_out) v = silu_f(v);
                y[(size_t)n * Out + o] = v;
            }
        }
    }
}

// ---- backward of linear_kernel (training path; M = batch <= 64 rows) --------------------------------------------------------------
// Both products run on the fp32 matrix core (v_mfma_f32_32x32x2f32, exact fp32 products, fp32 accumulate): they are skinny GEMMs
// whose one large operand -- the 51 MB FiLM projection weight of the LBBDM-f4 UNet, or its gradient -- moves once, at HBM rate.
//
// dx_pre[n][i] = act_in'(x[n][i]) * sum_o dy[n][o] w[o][i].  grid = (In/128, o-chunks); a workgroup stages its chunk of dy in LDS
// (pitch LB_PITCH: the A fragment's 64 lanes fall on 64 banks), each wave owns 32 input columns for all rows (two 32-row
// accumulators) and streams its slice of w with eight 128-byte row segments in flight per half-wave; the o-chunks write partials
// that a second kernel sums in a fixed order (deterministic).
constexpr int LB_OCH = 256;      // outputs per chunk
constexpr int LB_PITCH = LB_OCH + 2;

__global__ void __launch_bounds__(256) linear_bwd_x_partial_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                   float* __restrict__ part, int N, int In, int Out) {
    extern __shared__ __attribute__((aligned(16))) float dys[];              // [32 or 64][LB_PITCH]: dy[n][o in chunk]; rows >= N zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int o0 = blockIdx.y * LB_OCH;
    const int on = min(LB_OCH, Out - o0);
    const int NT = N > 32 ? 2 : 1;
    for (int k = tid; k < NT * 32 * LB_OCH; k += 256) {
        const int n = k / LB_OCH, o = k - n * LB_OCH;
        dys[n * LB_PITCH + o] = (n < N && o < on) ? dy[(size_t)n * Out + o0 + o] : 0.f;
    }
    __syncthreads();
    const int i0 = blockIdx.x * 128 + wave * 32;
    if (i0 >= In) return;
    const int i = i0 + l31;
    const bool iv = i < In;
    const float* wp = w + (size_t)o0 * In + (iv ? i : In - 1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const int on2 = (on + 1) & ~1;                                            // dys is zero past `on`; w rows are clamped below
    for (int k0 = 0; k0 < on2; k0 += 16) {
        float b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = k0 + 2 * u + hi;
            const float v = wp[(size_t)(o < on ? o : on - 1) * In];           // clamped address + select (no branch between loads)
            b[u] = (iv && o < on) ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = k0 + 2 * u + hi;
            if (k0 + 2 * u < on2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[l31 * LB_PITCH + o], b[u], acc0, 0, 0, 0);
                if (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dys[(32 + l31) * LB_PITCH + o], b[u], acc1, 0, 0, 0);
            }
        }
    }
    if (!iv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (n < N) part[((size_t)blockIdx.y * N + n) * In + i] = acc0[r];
        if (NT == 2 && 32 + n < N) part[((size_t)blockIdx.y * N + 32 + n) * In + i] = acc1[r];
    }
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

// dw[o][i] = sum_n dy[n][o] act_in(x[n][i]),  db[o] = sum_n dy[n][o].  act_in(x) is staged in LDS once per workgroup (rows padded
// with zeros to an even count); a workgroup walks 32-output tiles, its four waves take the 32-column input tiles in turn: the dy
// fragment of the tile (K = batch) stays in registers, one ds_read + one MFMA per two batch rows, rows of dw leave as 128-byte runs.
__global__ void __launch_bounds__(256) linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ db, int N, int In,
                                                           int Out, int act_in) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [N2][In], N2 = N rounded up to even
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int N2 = (N + 1) & ~1;
    for (int k = tid; k < N2 * In; k += 256) {
        float v = 0.f;
        if (k < N * In) {
            v = x[k];
            if (act_in) v = silu_f(v);
        }
        xs[k] = v;
    }
    __syncthreads();
    const int itiles = (In + 31) / 32, otiles = (Out + 31) / 32;
    for (int ot = blockIdx.x; ot < otiles; ot += gridDim.x) {
        const int o = ot * 32 + l31;
        const bool ov = o < Out;
        float a[32];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int n = 2 * kk + hi;
            a[kk] = (kk * 2 < N2 && n < N && ov) ? dy[(size_t)n * Out + o] : 0.f;
        }
        if (db && wave == 0 && hi == 0 && ov) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += dy[(size_t)n * Out + o];
            db[o] = s;
        }
        for (int it = wave; it < itiles; it += 4) {
            const int i = it * 32 + l31;
            const float* xp = xs + (i < In ? i : In - 1);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk)
                if (kk * 2 < N2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], xp[(2 * kk + hi) * In], acc, 0, 0, 0);
            if (i < In) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oo = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (oo < Out) dw[(size_t)oo * In + i] = acc[r];
                }
            }
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
    const size_t lds = (size_t)((N + 1) & ~1) * In * sizeof(float);
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
    const int otiles = cdiv(Out, 32);
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(otiles < 512 ? otiles : 512), dim3(256), lds, st, dy, x, dw, db, N, In, Out, act_in);
    if (dx) {
        const int chunks = cdiv(Out, LB_OCH);
        const size_t lds_x = (size_t)(N > 32 ? 64 : 32) * LB_PITCH * sizeof(float);
        static bool x_attr_dev[BBDM_MAX_DEVICES] = {};
        bool& x_attr = x_attr_dev[bbdm_device_slot()];
        if (lds_x > 64 * 1024 && !x_attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bwd_x_partial_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x) != hipSuccess) {
                bbdm_set_error("linear_bwd: hipFuncSetAttribute failed");
                return BBDM_E_LAUNCH;
            }
            x_attr = true;
        }
        hipLaunchKernelGGL(linear_bwd_x_partial_kernel, dim3(cdiv(In, 128), chunks), dim3(256), lds_x, st, dy, w, ws, N, In, Out);
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
    hipStream_t st = (hipStream_t)stream;
    {
        const int MB = N > 32 ? 2 : 1;
        const size_t lds_m = (size_t)MB * 32 * (In + 1) * sizeof(float);
        // (from 2048 outputs: below, the launch has too few 32-output tiles for the chip -- 512 -> 512 is 4 workgroups whose waves each
        // walk the whole K -- and the thread-per-(row, output) kernel below wins: 21.8 -> 12.9 us at batch 32, 16.1 -> 6.7 at batch 4)
        const int mfma_min_out = 2048;
        if (In % 8 == 0 && In >= 64 && Out >= mfma_min_out && (((uintptr_t)w & 15) == 0) && lds_m <= 160 * 1024) {
            static size_t lds_m_dev[BBDM_MAX_DEVICES][2] = {};
            size_t& have = lds_m_dev[bbdm_device_slot()][MB - 1];
            if (lds_m > 64 * 1024 && lds_m > have) {
                const void* f = MB == 2 ? reinterpret_cast<const void*>(linear_mfma_kernel<2>) : reinterpret_cast<const void*>(linear_mfma_kernel<1>);
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) != hipSuccess) {
                    bbdm_set_error("linear: hipFuncSetAttribute(%zu) failed", lds_m);
                    return BBDM_E_LAUNCH;
                }
                have = lds_m;
            }
            const int otiles = cdiv(Out, 32);
            const dim3 grid(cdiv(otiles, 4) < 256 ? cdiv(otiles, 4) : 256);
            if (MB == 2) hipLaunchKernelGGL(linear_mfma_kernel<2>, grid, dim3(256), lds_m, st, x, w, b, y, N, In, Out, act_in, act_out);
            else hipLaunchKernelGGL(linear_mfma_kernel<1>, grid, dim3(256), lds_m, st, x, w, b, y, N, In, Out, act_in, act_out);
            BBDM_CHECK_LAUNCH("linear(mfma)");
            return BBDM_OK;
        }
    }
    const size_t lds = (size_t)N * (In + 4) * sizeof(float);
    BBDM_REQUIRE(lds <= 160 * 1024, "linear: N*In too large for LDS staging (%zu B)", lds);
    const int NBl = ilog2(ceil_pow2(N));
    const int outs_per_block = 256 >> NBl;
    const bool vec = (In % 4 == 0) && (((uintptr_t)w & 15) == 0);
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

// bbdm_linear_f32 on weights packed once (inference: static weights).  packed = bbdm_linear_pack_f32(w) holds
// bbdm_linear_packed_bytes(Out, In) bytes; N <= 32 rows, In a multiple of 32 (<= 8192).  Same results bit for bit.
extern "C" size_t bbdm_linear_packed_bytes(int Out, int In) { return (size_t)((Out + 31) / 32) * 32 * (size_t)In * 4; }
extern "C" int bbdm_linear_packed_supported(int N, int In, int Out) {
    return N > 0 && N <= 32 && In > 0 && In % LP_KB == 0 && Out > 0 &&
           (size_t)32 * (In + 1) * 4 + 16 + (size_t)4 * LP_NS * LP_BLOCK <= 160 * 1024;
}
extern "C" int bbdm_linear_pack_f32(const float* w, void* packed, int Out, int In, void* stream) {
    BBDM_REQUIRE(w && packed && Out > 0 && In > 0 && In % LP_KB == 0 && (((uintptr_t)w | (uintptr_t)packed) & 15) == 0, "linear_pack: bad args");
    const int otiles = (Out + 31) / 32;
    const size_t total = (size_t)otiles * (In / LP_KB) * (LP_BLOCK / 16);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(linear_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)packed, Out, In, otiles);
    BBDM_CHECK_LAUNCH("linear_pack");
    return BBDM_OK;
}
extern "C" int bbdm_linear_packed_f32(const float* x, const void* packed, const float* b, float* y, int N, int In, int Out, int act_in,
                                      int act_out, void* stream) {
    BBDM_REQUIRE(x && packed && y, "linear_packed: null pointer");
    BBDM_REQUIRE(bbdm_linear_packed_supported(N, In, Out) && ((uintptr_t)packed & 15) == 0, "linear_packed: N=%d In=%d Out=%d unsupported",
                 N, In, Out);
    const size_t lds = (((size_t)32 * (In + 1) * 4 + 15) & ~(size_t)15) + (size_t)4 * LP_NS * LP_BLOCK;
    static size_t lds_dev[BBDM_MAX_DEVICES] = {};
    size_t& have = lds_dev[bbdm_device_slot()];
    if (lds > 64 * 1024 && lds > have) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_packed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            bbdm_set_error("linear_packed: hipFuncSetAttribute(%zu) failed", lds);
            return BBDM_E_LAUNCH;
        }
        have = lds;
    }
    const int otiles = (Out + 31) / 32;
    hipLaunchKernelGGL(linear_packed_kernel, dim3((unsigned)((otiles + 3) / 4)), dim3(256), lds, (hipStream_t)stream, x,
                       (const unsigned char*)packed, b, y, N, In, Out, act_in, act_out);
    BBDM_CHECK_LAUNCH("linear_packed");
    return BBDM_OK;
}
