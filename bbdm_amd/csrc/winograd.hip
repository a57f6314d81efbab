// winograd.hip -- 3x3 stride-1 convolution through Winograd F(m x m, 3x3), m = 2, 4, 6, on the fp32 matrix core.
//
// Same call sites as conv_igemm.hip (openaimodel.py:207,233,524,690 on the wide layers); the reference's cuDNN / MIOpen
// back ends make the same algorithmic choice for 3x3 convolutions (fp32 "Winograd non-fused" is F(4x4,3x3)).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per (m+2)x(m+2) input tile d -> m x m output tile Y, summed over channels
//
// (m+2)^2 multiplies per m^2 outputs instead of 9 m^2: the contraction drops from 18 P Cin Cout FLOP (P pixels) to
// 8 P Cin Cout (m = 2, 2.25x fewer) or 4.5 P Cin Cout (m = 4, 4x fewer).  Three launches:
//   (1) winograd_input_kernel : x NHWC -> V[(m+2)^2][tiles][Cin]                           (HBM-bound, adds only)
//   (2) conv_igemm_f32 in batched 1x1 mode: (m+2)^2 GEMMs  M_xi = V_xi [tiles x Cin] . U_xi [Cin x Cout]  (MFMA-bound)
//   (3) winograd_output_kernel: M[(m+2)^2][tiles][Cout] -> y NHWC (+ bias, + residual)     (HBM-bound, adds only)
// U_xi = G g G^T is precomputed once per weight update in the packed layout the GEMM wants.
// fp32 throughout.  Rounding error relative to an fp64 convolution (rms / max, Cin = 512, unit-variance activations):
// direct 2e-7 / 3e-7, m = 2: 5e-7 / 6e-7, m = 4: 3e-6 / 1e-5 -- all far inside the 1e-3 per-step bar.
// m = 6 (8x8 tiles, 64 transform points, 1.78 multiplies per output instead of 2.25; 6e-6 / 1.4e-5 in the same test,
// tests/test_winograd_math_cpu.py): H, W need not be multiples of 6 -- edge tiles read zeros beyond the image and their
// out-of-image outputs are not written.  Measured on MI355X (round 2, profiles/r02_wino_bench.txt): 1.1-1.25x faster
// than m = 4 per layer where the image is >= 64 pixels a side (little edge waste) and there are >= ~1000 tiles; slower
// on 16x16 / 32x32 latents, where m = 4 stays (bbdm_amd/unet.py: winograd_tile).
#include "winograd_math.h"
#include <stdlib.h>
#include "bf3_split.h"
#include "h2_split.h"
#include "stats_acc.h"

namespace {

constexpr int KC = 16;

// ---- (1) input transform: one thread = one (tile, channel quad) ------------------------------------------------------
// PRE: the tensor being convolved is act(x * sc[n][c] + bi[n][c]) (GroupNorm [+FiLM] [+SiLU] folded into per-image,
//      per-channel coefficients by bbdm_groupnorm_coeffs_f32); zero padding applies to the activated tensor.
// UP : x is [N, H/2, W/2] and is nearest-upsampled x2 on the fly (Upsample.forward, openaimodel.py:111-121).
template <int MO, bool PRE, bool UP>
__global__ void __launch_bounds__(256) winograd_input_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V,
                                                             const float* __restrict__ sc, const float* __restrict__ bi,
                                                             int pre_ld, int pre_silu, int N, int H, int W, int C,
                                                             size_t plane) {
    constexpr int AL = MO + 2;
    const int C4 = C >> 2;
    const int TH = H / MO, TW = W / MO;
    const long long total = (long long)N * TH * TW * C4;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256) {
        const int c = (int)(u % C4) * 4;
        const long long tile = u / C4;
        const int tw = (int)(tile % TW);
        const long long r = tile / TW;
        const int th = (int)(r % TH), n = (int)(r / TH);
        float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = f4zero();
        if (PRE) {
            s4 = *reinterpret_cast<const float4*>(sc + (size_t)n * pre_ld + c);
            b4 = *reinterpret_cast<const float4*>(bi + (size_t)n * pre_ld + c);
        }
        const int Hs = UP ? H >> 1 : H, Ws = UP ? W >> 1 : W;
        float4 t[AL][AL];                 // t[i][j] = (B^T d)[i][j]
#pragma unroll
        for (int j = 0; j < AL; ++j) {    // column j of the tile: its loads issued together (clamped addresses + 0/1 mask, see
            const int w = MO * tw - 1 + j;    // winograd_input6_kernel), then transformed down the rows
            const int wc = min(max(w, 0), W - 1);
            const int wsrc = UP ? wc >> 1 : wc;
            const float wmask = (w >= 0 && w < W) ? 1.f : 0.f;
            float4 d[AL], col[AL];
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int hc = min(max(MO * th - 1 + i, 0), H - 1);
                const int hs = UP ? hc >> 1 : hc;
                d[i] = *reinterpret_cast<const float4*>(x + ((size_t)(n * Hs + hs) * Ws + wsrc) * ldx + c);
            }
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int h = MO * th - 1 + i;
                const float mask = (h >= 0 && h < H) ? wmask : 0.f;
                float4 v = d[i];
                if (PRE) {
                    v.x = v.x * s4.x + b4.x; v.y = v.y * s4.y + b4.y;
                    v.z = v.z * s4.z + b4.z; v.w = v.w * s4.w + b4.w;
                    if (pre_silu) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
                }
                d[i] = mask * v;
            }
            bt_transform<MO>(d, col);
#pragma unroll
            for (int i = 0; i < AL; ++i) t[i][j] = col[i];
            __builtin_amdgcn_sched_barrier(0);          // keep the columns sequential: AL loads in flight, not AL^2
        }
        float* o = V + (size_t)tile * C + c;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            float4 row[AL];
            bt_transform<MO>(t[i], row);  // (B^T d B)[i][.] = B^T applied along the row
#pragma unroll
            for (int j = 0; j < AL; ++j) { *reinterpret_cast<float4*>(o) = row[j]; o += plane; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- (3) output transform: one thread = one (tile, output-channel quad) ---------------------------------------------
// GroupNorm statistics of the tensor being written, accumulated by its PRODUCER (openaimodel.py:205,229,306,688: every 3x3
// conv output of the UNet is normalised next): per (image, group) sum / sum of squares, up to two consumers with
// their own group width and channel offset (a block output is normalised by the next block as [C] and, through the concat,
// by an output block as a slice of [C + C']).  This removes the separate statistics pass -- a full re-read of every
// activation (25.8 GB and 5.3 ms per 256x256 / batch-16 step).  The sums are EXACT integer-limb accumulators (stats_acc.h): fp64
// partial sums of a thread are cut into limbs and added with integer atomics, in LDS and in HBM, so the result does not depend on
// the order in which waves and workgroups finish -- a sampling step is bitwise reproducible.  A workgroup owns a contiguous run of
// `iters` x 256 (tile, channel-group) units, i.e. a few consecutive tiles of at most ST_IMGS images: its partial sums meet in an
// LDS table and reach HBM as one cell (<= 3 integer atomics) per touched (image, group, sum | sum of squares).
struct StatArgs {
    unsigned long long* s[2];   // [N][32][2][SA_W] limb accumulators (zeroed by the caller), or null
    int cpg[2];                 // channels per group of that consumer (a multiple of the kernel's channel vector)
    int coff[2];                // channel offset of this tensor inside the consumer's tensor
};
constexpr int ST_IMGS = 4;
constexpr int ST_CELLS = 2 * ST_IMGS * 32 * 2;          // [consumer][image][group][sum | sq]
constexpr int ST_WORDS = ST_CELLS * SA_W;

__device__ __forceinline__ void stat_add(unsigned long long* lsum, const StatArgs& st, int n_local, int n, int c, double sum, double sq) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!st.s[k]) continue;
        const int g = (st.coff[k] + c) / st.cpg[k];
        if (n_local < ST_IMGS) {
            unsigned long long* d = lsum + ((size_t)((k * ST_IMGS + n_local) * 32 + g) * 2) * SA_W;
            sa_add(d, sum);
            sa_add(d + SA_W, sq);
        } else {                           // a run spanning more than ST_IMGS images (tiny images): straight to HBM
            unsigned long long* d = st.s[k] + ((size_t)(n * 32 + g) * 2) * SA_W;
            sa_add(d, sum);
            sa_add(d + SA_W, sq);
        }
    }
}
__device__ __forceinline__ void stat_flush(const unsigned long long* lsum, const StatArgs& st, int n0, int N, int nthreads = 256) {
    for (int i = threadIdx.x; i < ST_CELLS; i += nthreads) {
        const int k = i / (ST_IMGS * 64), rem = i - k * (ST_IMGS * 64);
        const int n = n0 + rem / 64;
        if (st.s[k] && n < N) sa_add_cell(st.s[k] + ((size_t)n * 64 + (rem & 63)) * SA_W, lsum + (size_t)i * SA_W);
    }
}

// y[MO*th + a][MO*tw + b] = (A^T m A)[a][b] + bias (+ residual)
template <int MO>
__global__ void __launch_bounds__(256) winograd_output_kernel(const float* __restrict__ M, size_t plane, int ldm, int splits,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res, int ldr, int res_per_image,
                                                              float* __restrict__ y, int ldy, int N, int H, int W, int Cout,
                                                              int iters, const StatArgs st, int ph) {
    constexpr int AL = MO + 2;          // ph: see winograd_output6_kernel
    __shared__ unsigned long long lsum[ST_WORDS];
    const int C4 = (ph ? 4 * Cout : Cout) >> 2;
    const int TH = H / MO, TW = W / MO;
    const long long total = (long long)N * TH * TW * C4;
    const long long base = (long long)blockIdx.x * iters * 256;
    const int n0 = (int)((base / C4) / ((long long)TH * TW));
    const bool stats = st.s[0] != nullptr || st.s[1] != nullptr;
    if (stats) {
        for (int i = threadIdx.x; i < ST_WORDS; i += 256) lsum[i] = 0ull;
        __syncthreads();
    }
    for (int it = 0; it < iters; ++it) {
        const long long u = base + (long long)it * 256 + threadIdx.x;
        if (u >= total) break;
        const int cm = (int)(u % C4) * 4;
        const int pq = ph ? cm / Cout : 0, c = cm - pq * Cout;
        const long long tile = u / C4;
        const int tw = (int)(tile % TW);
        const long long r = tile / TW;
        const int th = (int)(r % TH), n = (int)(r / TH);
        const float* m = M + (size_t)tile * ldm + cm;
        float4 s[MO][AL];                 // s = A^T m, built one column of m at a time
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            float4 v[AL], sj[MO];
#pragma unroll
            for (int i = 0; i < AL; ++i) v[i] = *reinterpret_cast<const float4*>(m + (size_t)(i * AL + j) * plane);
            for (int z = 1; z < splits; ++z)      // split-K tile GEMMs (small layers): partial sums M[z][xi][tiles][Cout], added in order
#pragma unroll
                for (int i = 0; i < AL; ++i)
                    v[i] = v[i] + *reinterpret_cast<const float4*>(m + ((size_t)z * (AL * AL) + (i * AL + j)) * plane);
            at_transform<MO>(v, sj);
#pragma unroll
            for (int a = 0; a < MO; ++a) s[a][j] = sj[a];
        }
        const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c) : f4zero();
        double psum = 0.0, psq = 0.0;
#pragma unroll
        for (int a = 0; a < MO; ++a) {
            float4 o[MO];
            at_transform<MO>(s[a], o);
            const size_t pix0 = (size_t)(n * H + MO * th + a) * W + MO * tw;
            float4 rv[MO];            // the residuals of one output row fetched together, then the row stored
#pragma unroll
            for (int b = 0; b < MO; ++b) rv[b] = f4zero();
            if (res) {
#pragma unroll
                for (int b = 0; b < MO; ++b)
                    rv[b] = *reinterpret_cast<const float4*>(
                        res_per_image == 1 ? res + (size_t)n * ldr + c
                        : res_per_image == 2 ? res + ((size_t)(n * (H >> 1) + ((MO * th + a) >> 1)) * (W >> 1) + ((MO * tw + b) >> 1)) * ldr + c
                                             : res + (pix0 + b) * ldr + c);
            }
#pragma unroll
            for (int b = 0; b < MO; ++b) {
                const float4 val = (o[b] + b4) + rv[b];
                const size_t pix = ph ? (size_t)(n * 2 * H + 2 * (MO * th + a) + (pq >> 1)) * (2 * W) + 2 * (MO * tw + b) + (pq & 1)
                                      : pix0 + b;
                *reinterpret_cast<float4*>(y + pix * ldy + c) = val;
                if (stats) {
                    psum += ((double)val.x + (double)val.y) + ((double)val.z + (double)val.w);
                    psq += ((double)val.x * val.x + (double)val.y * val.y) + ((double)val.z * val.z + (double)val.w * val.w);
                }
            }
        }
        if (stats) stat_add(lsum, st, n - n0, n, c, psum, psq);
    }
    if (stats) {
        __syncthreads();
        stat_flush(lsum, st, n0, N);
    }
}

// ---- m = 6 (see the header): 8x8 tiles, one thread = one (tile, channel PAIR) so that the 64 values of
// a tile stay in registers; TH / TW are rounded up and edge tiles are masked.
template <bool PRE, bool UP>
__global__ void __launch_bounds__(256) winograd_input6_kernel(const float* __restrict__ x, int ldx, float* __restrict__ V,
                                                              const float* __restrict__ sc, const float* __restrict__ bi,
                                                              int pre_ld, int pre_silu, int N, int H, int W, int C,
                                                              size_t plane) {
    constexpr int MO = 6, AL = 8;
    const int C2 = C >> 1;
    const int TH = (H + MO - 1) / MO, TW = (W + MO - 1) / MO;
    const long long total = (long long)N * TH * TW * C2;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256) {
        const int c = (int)(u % C2) * 2;
        const long long tile = u / C2;
        const int tw = (int)(tile % TW);
        const long long r = tile / TW;
        const int th = (int)(r % TH), n = (int)(r / TH);
        float2 s2 = make_float2(1.f, 1.f), b2 = make_float2(0.f, 0.f);
        if (PRE) {
            s2 = *reinterpret_cast<const float2*>(sc + (size_t)n * pre_ld + c);
            b2 = *reinterpret_cast<const float2*>(bi + (size_t)n * pre_ld + c);
        }
        const int Hs = UP ? H >> 1 : H, Ws = UP ? W >> 1 : W;
        float2 t[AL][AL];
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            // One column of the window at a time, its 8 loads issued back to back: out-of-image taps read a clamped (valid)
            // address and are zeroed by a 0/1 factor afterwards.  With a bounds-check branch per tap the waitcnt pass falls
            // back to vmcnt(0) between the loads -- ONE load in flight per thread, which is what held this HBM-bound kernel at
            // 2.9 TB/s.  (Batching the whole 8x8 window instead costs 128 more VGPRs: the m = 4 experiment of that kind lost.)
            const int w = MO * tw - 1 + j;
            const int wc = min(max(w, 0), W - 1);
            const int wsrc = UP ? wc >> 1 : wc;
            const float wmask = (w >= 0 && w < W) ? 1.f : 0.f;
            float2 d[AL], col[AL];
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int hc = min(max(MO * th - 1 + i, 0), H - 1);
                const int hs = UP ? hc >> 1 : hc;
                d[i] = *reinterpret_cast<const float2*>(x + ((size_t)(n * Hs + hs) * Ws + wsrc) * ldx + c);
            }
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int h = MO * th - 1 + i;
                const float mask = (h >= 0 && h < H) ? wmask : 0.f;
                float2 v = d[i];
                if (PRE) {
                    v.x = v.x * s2.x + b2.x; v.y = v.y * s2.y + b2.y;
                    if (pre_silu) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); }
                }
                d[i] = make_float2(mask * v.x, mask * v.y);
            }
            bt_transform<MO>(d, col);
#pragma unroll
            for (int i = 0; i < AL; ++i) t[i][j] = col[i];
            __builtin_amdgcn_sched_barrier(0);          // keep the columns sequential: 8 loads in flight, not 64
        }
        float* o = V + (size_t)tile * C + c;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            float2 row[AL];
            bt_transform<MO>(t[i], row);
#pragma unroll
            for (int j = 0; j < AL; ++j) { *reinterpret_cast<float2*>(o) = row[j]; o += plane; }
            __builtin_amdgcn_sched_barrier(0);      // rows sequential, ONE walking pointer: the 64 plane addresses computed up front
                                                    // cost 2 VGPRs each (non-fused variant: 472 VGPRs = 1 wave per SIMD; now 226)
        }
    }
}

// ---- (1') input transform that writes the three bf16 planes gemm_bf3p.hip consumes -----------------------------------------
// Same arithmetic as the kernels above (B^T d B in fp32, GroupNorm -> FiLM -> SiLU / nearest x2 folded in), followed by the
// exact three-way bf16 split of every transformed value (bf3_split.h): the GEMM's main loop then copies and multiplies, nothing
// else.  Output = the A-plane layout of gemm_bf3p.hip, [xi][tile / 32][chunk][3][1 KB fragment unit].  (The first version -- one
// thread per (tile, channel pair) holding the whole window: 174 VGPRs at m = 6 -- was removed in round 5; the two-phase kernel
// below replaced it in round 3.)

// The lanes of a wave hold, for AL transform points xi = (i, 0 .. AL-1) and the three bf16 planes, one dword each = the channel pair
// 2 cp, 2 cp + 1 of tile tl (lane = tl * 8 + cp: 8 consecutive tiles x 16 channels).  Write them TRANSPOSED: per (xi, plane) the
// [8 tiles][16 channels] block becomes [16 channels][8 tiles] -- 16-byte runs of 8 consecutive tiles per channel, 256 B contiguous --
// inside the fragment unit (rows = channels c % 32, k = tiles t % 16) of the layout [c / 32][t / 16][3][1 KB].  The 2-byte transpose is the
// LDS transpose read: the block is staged row-major ([tile][16 ch] = 32-B rows) in the wave's own scratch and a 16-lane group reads
// it back with ds_read_b64_tr_b16 -- lane i of the group receives channel i, 4 tiles per read.  Four groups = four (xi, plane)
// blocks per pass; AL / 2 transform points per round so that the scratch stays inside the wave's 8 x 64 x 8 B region.
template <int AL, int NPL = 3>
__device__ __forceinline__ void store_transposed(const unsigned (&pl)[AL][NPL], unsigned char* scratch, int lane, unsigned char* dst_xi0,
                                                 size_t plane_t, int chunk, int tile0, int tchunks) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    constexpr int HALF = AL / 2, COMBOS = NPL * HALF, PITCH = 256 + 64;     // bytes between blocks (+64: stagger the bank rows)
    static_assert(COMBOS * PITCH <= AL * 64 * 8, "transposition scratch exceeds the wave's LDS region");
    const int grp = lane >> 4, i16 = lane & 15;
    const int c = chunk * KC + i16;                                           // this lane's channel in the transposed store
    // unit (c / 32, t / 16), k-half (t % 16) / 8, row c % 32
    unsigned char* dst = dst_xi0 + (((size_t)(c >> 5) * tchunks + (tile0 >> 4)) * NPL) * 1024 + ((tile0 >> 3) & 1) * 512 + (c & 31) * 16;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        // (the wave runs in lockstep: every lane has read its row of the intermediate / the previous round's blocks before this
        // store instruction issues; the wave barrier only pins that order for the compiler -- and for the CPU emulation's fibers)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < HALF; ++q)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                *reinterpret_cast<unsigned*>(scratch + (q * NPL + p) * PITCH + lane * 4) = pl[round * HALF + q][p];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < (COMBOS + 3) / 4; ++it) {
            const int combo = it * 4 + grp;
            const int cc = combo < COMBOS ? combo : COMBOS - 1;       // (every lane reads: no divergence around the cross-lane read)
            const unsigned char* a = scratch + cc * PITCH + (i16 >> 2) * 32 + (i16 & 3) * 8;
            const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a));
            const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a + 4 * 32));
            typedef short v8s __attribute__((ext_vector_type(8)));
            const v8s both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const int q = cc / NPL, p = cc - NPL * q;
            if (combo < COMBOS)
                *reinterpret_cast<uint4*>(dst + (size_t)(round * HALF + q) * plane_t + p * 1024) = __builtin_bit_cast(uint4, both);
        }
    }
}

// ---- (1'') the same transform in two phases through LDS: (m+2) waves per workgroup, 8x the threads in flight -------------------------
// The one-thread-per-(tile, channel pair) kernel above keeps the whole (m+2)^2 window of its pair in registers: 174 VGPRs at m = 6,
// two waves per SIMD -- and its ablation (tools/wino_variants.py: loads alone 0.145 ms, stores alone 0.159 ms, together 0.258 ms at
// 64x64 x 1024 channels) shows the two memory streams and the arithmetic overlapping poorly with so few waves.  Here a workgroup owns
// 64 units (8 consecutive tiles x the 8 channel pairs of one 16-channel chunk), lane = unit, and its m + 2 waves split the window:
//   phase A: wave j loads column j of the window (m + 2 loads per lane, issued together), applies the fused GroupNorm -> FiLM ->
//            SiLU, transforms it down the rows (B^T d) and leaves the m + 2 results in LDS [i][j][unit];
//   phase B: wave i reads row i of the intermediate ([i][0..m+1][unit]: consecutive lanes, consecutive 8-byte words), transforms it
//            (. B), splits each value into its three bf16 planes and stores them -- per plane and transform point the wave writes
//            8 rows x 16 B of the fragment unit twice = two full 128-B lines, as above.
// ~60 VGPRs and 32 KB of LDS per workgroup: 4 workgroups = 32 waves per CU.
// TR (training forward of a layer whose weight gradient is taken in the Winograd domain): the planes are ALSO written transposed,
// Vt[xi][ci / 32][tile / 16][3][1 KB unit: rows = 32 channels, k = 16 tiles] -- the A operand of dU_xi = V_xi^T dM_xi as the SAME
// gemm_bf3p kernel takes it (rows = ci, contraction over the tiles): see store_transposed().
// What bounds it (round 3, profiles/r03_transform_bound.md): HBM, at the rate this chip gives a stream that is 73 % WRITES.  A
// bare streaming kernel with one read and three write streams moves 3.5 - 5.2 TB/s (tools/microbench/hbm_mix.hip; a pure read stream
// 6.3 - 6.5, a copy 4.9 - 5.7); this kernel moves 5.0.  Checked and ruled out: the VALU (the index arithmetic below cut its issue
// slots per wave from ~630 to ~480: -3 %), load latency (a variant that walked 2 - 16 chunks per workgroup and requested chunk
// k + 1's rows before transforming chunk k was 5 - 7 % SLOWER: fewer resident waves, nothing to hide), the `nt` store policy
// (profiles/r03_nt_and_output_lds_ab.txt).  Fewer bytes is the only lever left: 6 B per transformed element is what the exact
// three-way split costs.
// Index arithmetic: the tile's (image, row, column) comes from two multiply-high divisions by launch constants (FastDiv, common.h)
// instead of two 64-bit divisions by kernel arguments; a row address is ONE 24-bit multiply-add on a 32-bit element index (IDX64:
// tensors of 2^32 elements or more, or a row pitch of 2^24 elements or more, keep 64-bit row addresses); SiLU evaluates a channel
// pair with packed multiplies / adds around its two v_exp_f32 / v_rcp_f32.
// GNC (round 4, small problems only): the kernel forms the fused-producer coefficients ITSELF from the GroupNorm statistics, gamma,
// beta and the FiLM vector -- the expressions of gn_coeffs_kernel (groupnorm.hip), hence the same bits -- instead of reading sc / bi
// that a separate launch wrote: one launch fewer per fused GroupNorm (41 per forward, ~6 us each where every launch is ~10 us).
// Every thread folds the limbs of its (image, group) and does the fp64 division / square root for its channel pair: free where the
// launch is latency-bound, not where the transform is HBM-bound with tens of thousands of workgroups (measured in round 4 with a
// table of (mean, rstd): +2.3 ms on the C2 step) -- the caller chooses (unet.py: BBDM_GN_IN_TRANSFORM).
struct GnFold {
    const unsigned long long* stats;      // [N][G][2][SA_W] limbs (stats_acc.h)
    const float* gamma;                   // [C]
    const float* beta;                    // [C]
    const float* film;                    // [N][film_ld]: scale at c, shift at C + c; or null
    int film_ld, C, G, cpg;               // cpg = C / G, even
    double cnt;                           // values per (image, group): HW * cpg
    float eps;
};
// F32 (round 5): the same two phases, the transformed values stored as fp32 rows V[xi][tile][Cin] (4 B per element instead of the
// planes' 6) for the layers whose tile GEMM is HBM-bound (Cout = 128 at the 256^2 level: 10 B of operands per 256 FLOP) -- there
// gemm_bf3.hip splits V under its own idle matrix pipe and both launches move a third fewer operand bytes.  Per transform point a wave
// writes 8 tiles x 64 B; the other 16-channel chunks of the tile group run on the same XCD (see below) and complete the lines in its L2.
// ST (round 5): the tile stride.  ST = 7 with MO = 6 is the input side of F(7x7, 2x2) (the phase filters of an up-sampling conv,
// winograd_math.h): the same 8 x 8 window and B^T, windows 7 pixels apart (window rows 7 t - 1 .. 7 t + 6).
// NPL (round 6): planes per value.  3 = the exact bf16 split; 2 = the fp16 pair of h2_split.h under the scale 2^e, e =
// h2_exp_of_bound(*hbound x gain) -- hbound: a device float >= max |d| over the whole transformed tensor (after the fused producer:
// groupnorm.hip, h2_gn_bounds_kernel), gain = wino_input_gain: |B^T d B| <= gain max |d| -- 4 B per transformed element instead of 6,
// units [..][nchunks][2][1 KB].
template <int MO, bool PRE, bool UP, bool TR, bool IDX64, bool GNC = false, bool F32 = false, int ST = MO, int NPL = 3>
__global__ void __launch_bounds__((MO + 2) * 64) winograd_input_split2_kernel(const float* __restrict__ x, int ldx,
                                                                              unsigned char* __restrict__ Vp,
                                                                              const float* __restrict__ sc, const float* __restrict__ bi,
                                                                              int pre_ld, int pre_silu, int N, int H, int W, int nchunks,
                                                                              unsigned T, int TG, size_t plane,
                                                                              unsigned char* __restrict__ Vt, size_t plane_t,
                                                                              int tchunks, const FastDiv dTW, const FastDiv dTH,
                                                                              const FastDiv dCH, const GnFold gn,
                                                                              const float* __restrict__ hbound, int vt_h2 = 0) {
    static_assert(NPL == 3 || (NPL == 2 && !F32), "the fp16-pair planes have no fp32 form");
    constexpr int AL = MO + 2;
    __shared__ float2 lds[AL * AL * 64];
    const unsigned L = blockIdx.x, q = L >> 3;
    // the chunks of one tile group run on ONE XCD (block id % 8): the two 64-B halves of an input line meet in that L2
    const unsigned qc = fastdiv(q, dCH);
    const int chunk = (int)(q - qc * (unsigned)nchunks), tg = (int)(qc * 8 + (L & 7));
    if (tg >= TG) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int TH = (int)dTH.d, TW = (int)dTW.d;
    // lane = tile x channel pair: 8 tiles x 8 pairs (one 16-channel chunk); F32: 4 tiles x 16 pairs (32 channels), so that a tile's row of
    // the fp32 V is one full 128-B line per load and per store (64-B half lines at a pitch of 4 Cin bytes held the 640-channel layer at
    // 3.5 TB/s); nchunks / TG / tg / chunk then count 32-channel chunks and groups of four tiles (the launcher's grid follows)
    constexpr int TPG = F32 ? 4 : 8, CPW = F32 ? 16 : 8;
    const int tl = lane / CPW, cp = lane % CPW;
    const unsigned tile = (unsigned)tg * TPG + tl;
    const int c = chunk * (2 * CPW) + cp * 2;
    {   // ---- phase A: column `wave` of the window ------------------------------------------------------------------------------
        const int jj = wave;
        float2 d[AL], col[AL];
        if (tile < T) {
            const unsigned r = fastdiv(tile, dTW), n = fastdiv(r, dTH);
            const int tw = (int)(tile - r * (unsigned)TW), th = (int)(r - n * (unsigned)TH);
            float2 s2 = make_float2(1.f, 1.f), b2 = make_float2(0.f, 0.f);
            if (PRE && GNC) {
                // gn_coeffs_kernel's expressions, for channels c and c + 1 (same group: cpg is even)
                const unsigned long long* cell = gn.stats + (((size_t)n * gn.G + c / gn.cpg) * 2) * SA_W;
                const double s = sa_load(cell), ss = sa_load(cell + SA_W);
                const double mean = s / gn.cnt;
                double var = ss / gn.cnt - mean * mean;
                var = var > 0.0 ? var : 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)gn.eps));
                const float fmean = (float)mean;
                const float2 ga = *reinterpret_cast<const float2*>(gn.gamma + c), be = *reinterpret_cast<const float2*>(gn.beta + c);
                float sc0 = rstd * ga.x, sc1 = rstd * ga.y;
                float bi0 = be.x - fmean * sc0, bi1 = be.y - fmean * sc1;
                if (gn.film) {
                    const float2 fs = *reinterpret_cast<const float2*>(gn.film + (size_t)n * gn.film_ld + c);
                    const float2 fb = *reinterpret_cast<const float2*>(gn.film + (size_t)n * gn.film_ld + gn.C + c);
                    sc0 = sc0 * (1.f + fs.x); sc1 = sc1 * (1.f + fs.y);
                    bi0 = bi0 * (1.f + fs.x) + fb.x; bi1 = bi1 * (1.f + fs.y) + fb.y;
                }
                s2 = make_float2(sc0, sc1);
                b2 = make_float2(bi0, bi1);
            } else if (PRE) {
                s2 = *reinterpret_cast<const float2*>(sc + (size_t)n * pre_ld + c);
                b2 = *reinterpret_cast<const float2*>(bi + (size_t)n * pre_ld + c);
            }
            const int Hs = UP ? H >> 1 : H, Ws = UP ? W >> 1 : W;
            const int w = ST * tw - 1 + jj;
            const int wc = min(max(w, 0), W - 1);
            const int wsrc = UP ? wc >> 1 : wc;
            const float wmask = (w >= 0 && w < W) ? 1.f : 0.f;
            const int h0 = ST * th - 1;
            if (IDX64) {
#pragma unroll
                for (int i = 0; i < AL; ++i) {        // clamped addresses, out-of-image taps zeroed afterwards: the loads issue together
                    const int hc = min(max(h0 + i, 0), H - 1);
                    const int hs = UP ? hc >> 1 : hc;
                    d[i] = *reinterpret_cast<const float2*>(x + ((size_t)(n * Hs + hs) * Ws + wsrc) * ldx + c);
                }
            } else {
                const unsigned rowstride = (unsigned)Ws * (unsigned)ldx;                     // < 2^24 (host check)
                const unsigned e0 = (n * (unsigned)Hs * (unsigned)Ws + (unsigned)wsrc) * (unsigned)ldx + (unsigned)c;
#pragma unroll
                for (int i = 0; i < AL; ++i) {
                    const int hc = min(max(h0 + i, 0), H - 1);
                    const unsigned hs = (unsigned)(UP ? hc >> 1 : hc);
                    d[i] = *reinterpret_cast<const float2*>(x + (e0 + __umul24(hs, rowstride)));
                }
            }
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int h = h0 + i;
                const float mask = (h >= 0 && h < H) ? wmask : 0.f;
                float2 v = d[i];
                if (PRE) {
                    v.x = v.x * s2.x + b2.x; v.y = v.y * s2.y + b2.y;
                    if (pre_silu) v = silu_fast2(v);
                }
                d[i] = make_float2(mask * v.x, mask * v.y);
            }
        } else {                                      // rows between the real and the padded tile count: zeros
#pragma unroll
            for (int i = 0; i < AL; ++i) d[i] = make_float2(0.f, 0.f);
        }
        bt_transform<MO>(d, col);
#pragma unroll
        for (int i = 0; i < AL; ++i) lds[(i * AL + jj) * 64 + lane] = col[i];
    }
    __syncthreads();
    {   // ---- phase B: row `wave` of the intermediate -> the 3 bf16 planes of transform points (wave, 0 .. m + 1) ------------------
        const int i = wave;
        float2 t[AL], row[AL];
#pragma unroll
        for (int jj = 0; jj < AL; ++jj) t[jj] = lds[(i * AL + jj) * 64 + lane];
        bt_transform<MO>(t, row);
        if (F32) {
            float* of = reinterpret_cast<float*>(Vp + (size_t)(i * AL) * plane) + (size_t)tile * ((size_t)nchunks * (2 * CPW)) + c;
#pragma unroll
            for (int jj = 0; jj < AL; ++jj) {
                *reinterpret_cast<float2*>(of) = row[jj];
                of += plane >> 2;
            }
            return;
        }
        const int g = (int)(tile >> 5), rl = (int)(tile & 31);
        if constexpr (NPL == 2) {
            const float hs = h2_pow2(h2_exp_of_bound(*hbound * wino_input_gain(ST == 7 ? 7 : MO)));
            unsigned char* o = Vp + (size_t)(i * AL) * plane + ((size_t)g * nchunks + chunk) * 2 * 1024 + (cp >> 2) * 512 + rl * 16 + (cp & 3) * 4;
#pragma unroll
            for (int jj = 0; jj < AL; ++jj) {
                unsigned p1, p2;
                h2_split2(row[jj].x * hs, row[jj].y * hs, p1, p2);
                *reinterpret_cast<unsigned*>(o) = p1;
                *reinterpret_cast<unsigned*>(o + 1024) = p2;
                o += plane;
            }
            if constexpr (TR) {
                // training forward: the forward GEMM reads the fp16-pair planes above; the weight gradient contracts the TRANSPOSED copy
                // with dM.  vt_h2 (UNetModel.gemm_h2_train = 3): that GEMM runs on the fp16 pair as well (dM under the measured maximum
                // of dY) and the copy holds the same two planes, transposed; else it stays on the exact bf16 split, and so does the copy
                if (vt_h2) {
                    unsigned pl[AL][2];
#pragma unroll
                    for (int jj = 0; jj < AL; ++jj) h2_split2(row[jj].x * hs, row[jj].y * hs, pl[jj][0], pl[jj][1]);
                    store_transposed<AL, 2>(pl, reinterpret_cast<unsigned char*>(lds + (size_t)i * AL * 64), lane,
                                            Vt + (size_t)(i * AL) * plane_t, plane_t, chunk, (int)(tile - tl), tchunks);
                } else {
                    unsigned pl[AL][3];
#pragma unroll
                    for (int jj = 0; jj < AL; ++jj) split2(row[jj].x, row[jj].y, pl[jj][0], pl[jj][1], pl[jj][2]);
                    store_transposed<AL>(pl, reinterpret_cast<unsigned char*>(lds + (size_t)i * AL * 64), lane,
                                         Vt + (size_t)(i * AL) * plane_t, plane_t, chunk, (int)(tile - tl), tchunks);
                }
            }
            return;
        }
        // byte (k >> 3) * 512 + r * 16 + (k & 7) * 2 of the unit, k = 2 cp
        unsigned char* o = Vp + (size_t)(i * AL) * plane + ((size_t)g * nchunks + chunk) * 3 * 1024 + (cp >> 2) * 512 + rl * 16 + (cp & 3) * 4;
        unsigned pl[AL][3];
#pragma unroll
        for (int jj = 0; jj < AL; ++jj) {
            split2(row[jj].x, row[jj].y, pl[jj][0], pl[jj][1], pl[jj][2]);
#if BBDM_NT_VSTORE
            store_nt(reinterpret_cast<unsigned*>(o), pl[jj][0]);
            store_nt(reinterpret_cast<unsigned*>(o + 1024), pl[jj][1]);
            store_nt(reinterpret_cast<unsigned*>(o + 2048), pl[jj][2]);
#else
            *reinterpret_cast<unsigned*>(o) = pl[jj][0];
            *reinterpret_cast<unsigned*>(o + 1024) = pl[jj][1];
            *reinterpret_cast<unsigned*>(o + 2048) = pl[jj][2];
#endif
            o += plane;
        }
        if (TR)     // row i of the intermediate has been consumed: its LDS region (this wave's alone from here on) is the scratch
            store_transposed<AL>(pl, reinterpret_cast<unsigned char*>(lds + (size_t)i * AL * 64), lane,
                                 Vt + (size_t)(i * AL) * plane_t, plane_t, chunk, (int)(tile - tl), tchunks);
    }
}

// ---- dM = A dY A^T for the Winograd-domain weight gradient, written as TRANSPOSED bf16 planes ----------------------------------
// dMt[xi][co / 32][tile / 16][3][1 KB unit: rows = 32 output channels, k = 16 tiles] = the B operand of dU_xi = V_xi^T dM_xi on the
// gemm_bf3p kernel (contraction over the tiles), in two phases through LDS like the input transform above: wave j < m transforms
// column j of the m x m window of dY (A d), wave i < m + 2 row i (. A^T), splits, and stores through store_transposed().  The one
// fp32 plane the bias gradient needs -- xi = (1, 1): row 1 of A is all ones, so it holds the tile sums of dY -- goes to dm11[tile][C].
// m = 8 has no transform point 1 (winograd_math.h): there the waves of phase A leave their column sums in LDS and wave 1 adds them in
// column order -- dm11 holds the tile sums of dY for every m.
// NPL = 2 (round 6): the fp16 pair under the scale of dybound x wino_dy_gain(m) (dybound: the measured maximum of dY).
template <int MO, int NPL = 3>
__global__ void __launch_bounds__((MO + 2) * 64) winograd_dy_split_kernel(const float* __restrict__ dy, int ld,
                                                                          unsigned char* __restrict__ dMt, float* __restrict__ dm11,
                                                                          int N, int H, int W, int C, int nchunks, long long T, int TG,
                                                                          size_t plane_t, int tchunks, const float* __restrict__ dybound) {
    constexpr int AL = MO + 2;
    __shared__ float2 lds[AL * AL * 64];                 // [i][j < m][unit] intermediates; later each wave's transposition scratch
    __shared__ float2 csum[MO == 8 ? MO * 64 : 1];       // m = 8: the column sums of the window (the bias gradient's tile sums)
    const int L = (int)blockIdx.x, q = L >> 3;
    const int chunk = q % nchunks, tg = (q / nchunks) * 8 + (L & 7);
    if (tg >= TG) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int TH = (H + MO - 1) / MO, TW = (W + MO - 1) / MO;
    const int tl = lane >> 3, cp = lane & 7;
    const long long tile = (long long)tg * 8 + tl;
    const int c = chunk * KC + cp * 2;
    if (wave < MO) {
        const int jj = wave;
        float2 d[MO], col[AL];
        if (tile < T && c < C) {
            const int tw = (int)(tile % TW);
            const long long r = tile / TW;
            const int th = (int)(r % TH), n = (int)(r / TH);
            const int w = MO * tw + jj;
            const int wc = min(w, W - 1);
            const float wmask = w < W ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < MO; ++i) {
                const int hc = min(MO * th + i, H - 1);
                d[i] = *reinterpret_cast<const float2*>(dy + ((size_t)(n * H + hc) * W + wc) * ld + c);
            }
#pragma unroll
            for (int i = 0; i < MO; ++i) d[i] = ((MO * th + i < H) ? wmask : 0.f) * d[i];
        } else {                                          // padded tiles / channels: zeros (they enter the contraction over the tiles)
#pragma unroll
            for (int i = 0; i < MO; ++i) d[i] = make_float2(0.f, 0.f);
        }
        a_transform<MO>(d, col);
#pragma unroll
        for (int i = 0; i < AL; ++i) lds[(i * AL + jj) * 64 + lane] = col[i];
        if constexpr (MO == 8) {
            float2 cs = d[0];
#pragma unroll
            for (int i = 1; i < MO; ++i) cs = cs + d[i];
            csum[jj * 64 + lane] = cs;
        }
    }
    __syncthreads();
    {
        const int i = wave;
        float2 t[MO], row[AL];
#pragma unroll
        for (int jj = 0; jj < MO; ++jj) t[jj] = lds[(i * AL + jj) * 64 + lane];
        a_transform<MO>(t, row);
        if constexpr (MO == 8) {
            if (i == 1 && tile < T && c < C) {
                float2 ts = csum[lane];
#pragma unroll
                for (int jj = 1; jj < MO; ++jj) ts = ts + csum[jj * 64 + lane];
                *reinterpret_cast<float2*>(dm11 + (size_t)tile * C + c) = ts;
            }
        } else {
            if (i == 1 && tile < T && c < C) *reinterpret_cast<float2*>(dm11 + (size_t)tile * C + c) = row[1];
        }
        if constexpr (NPL == 2) {
            const float hs = h2_pow2(h2_exp_of_bound(*dybound * wino_dy_gain(MO)));
            unsigned pl[AL][2];
#pragma unroll
            for (int jj = 0; jj < AL; ++jj) h2_split2(row[jj].x * hs, row[jj].y * hs, pl[jj][0], pl[jj][1]);
            store_transposed<AL, 2>(pl, reinterpret_cast<unsigned char*>(lds + (size_t)i * AL * 64), lane, dMt + (size_t)(i * AL) * plane_t,
                                    plane_t, chunk, (int)(tile - tl), tchunks);
        } else {
            unsigned pl[AL][3];
#pragma unroll
            for (int jj = 0; jj < AL; ++jj) split2(row[jj].x, row[jj].y, pl[jj][0], pl[jj][1], pl[jj][2]);
            store_transposed<AL>(pl, reinterpret_cast<unsigned char*>(lds + (size_t)i * AL * 64), lane, dMt + (size_t)(i * AL) * plane_t,
                                 plane_t, chunk, (int)(tile - tl), tchunks);
        }
    }
}

template <bool RES>
__global__ void __launch_bounds__(256) winograd_output6_kernel(const float* __restrict__ M, size_t plane, int ldm,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ res, int ldr, int res_per_image,
                                                               float* __restrict__ y, int ldy, int N, int H, int W, int Cout,
                                                               int iters, const StatArgs st, int ph) {
    // ph (BBDM_CONV_OUT_PHASES): M carries 4 Cout channels, channel (2 pa + pb) Cout + co of the tile grid position (oh, ow) is the
    // output pixel (2 oh + pa, 2 ow + pb), channel co, of a [N, 2H, 2W] image (the four phase filters of conv3x3(nearest x2(x)))
    constexpr int MO = 6, AL = 8;
    __shared__ unsigned long long lsum[ST_WORDS];
    const int C2 = (ph ? 4 * Cout : Cout) >> 1;
    const int TH = (H + MO - 1) / MO, TW = (W + MO - 1) / MO;
    const long long total = (long long)N * TH * TW * C2;
    const long long base = (long long)blockIdx.x * iters * 256;
    const int n0 = (int)((base / C2) / ((long long)TH * TW));
    const bool stats = st.s[0] != nullptr || st.s[1] != nullptr;
    if (stats) {
        for (int i = threadIdx.x; i < ST_WORDS; i += 256) lsum[i] = 0ull;
        __syncthreads();
    }
    for (int it = 0; it < iters; ++it) {
        const long long u = base + (long long)it * 256 + threadIdx.x;
        if (u >= total) break;
        const int cm = (int)(u % C2) * 2;
        const int pq = ph ? cm / Cout : 0, c = cm - pq * Cout;
        const long long tile = u / C2;
        const int tw = (int)(tile % TW);
        const long long r = tile / TW;
        const int th = (int)(r % TH), n = (int)(r / TH);
        const float* m = M + (size_t)tile * ldm + cm;
        float2 s[MO][AL];
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            float2 v[AL], sj[MO];
#pragma unroll
#if BBDM_NT_MLOAD
            for (int i = 0; i < AL; ++i) v[i] = load_nt(reinterpret_cast<const float2*>(m + (size_t)(i * AL + j) * plane));
#else
            for (int i = 0; i < AL; ++i) v[i] = *reinterpret_cast<const float2*>(m + (size_t)(i * AL + j) * plane);
#endif
            at_transform<MO>(v, sj);
#pragma unroll
            for (int a = 0; a < MO; ++a) s[a][j] = sj[a];
        }
        const float2 b2 = bias ? *reinterpret_cast<const float2*>(bias + c) : make_float2(0.f, 0.f);
        double psum = 0.0, psq = 0.0;
#pragma unroll
        for (int a = 0; a < MO; ++a) {
            float2 o[MO];
            at_transform<MO>(s[a], o);
            const int oh = MO * th + a;
            const int ohc = min(oh, H - 1);
            // the residuals of one output row are fetched together (clamped addresses for the masked edge pixels), then the
            // row is stored: with a load inside each pixel's bounds branch every store waited for the previous pixel's
            // round trip (gfx9 counts loads and stores in the one in-order vmcnt)
            float2 rv[MO];
            if (RES) {
#pragma unroll
                for (int b = 0; b < MO; ++b) {
                    const int owc = min(MO * tw + b, W - 1);
                    const float* rp = res_per_image == 1 ? res + (size_t)n * ldr + c
                                      : res_per_image == 2 ? res + ((size_t)(n * (H >> 1) + (ohc >> 1)) * (W >> 1) + (owc >> 1)) * ldr + c
                                                           : res + ((size_t)(n * H + ohc) * W + owc) * ldr + c;
                    rv[b] = *reinterpret_cast<const float2*>(rp);
                }
            }
#pragma unroll
            for (int b = 0; b < MO; ++b) {
                const int ow = MO * tw + b;
                if (oh < H && ow < W) {
                    float2 val = o[b] + b2;
                    if (RES) val = val + rv[b];
                    const size_t pix = ph ? (size_t)(n * 2 * H + 2 * oh + (pq >> 1)) * (2 * W) + 2 * ow + (pq & 1)
                                          : (size_t)(n * H + oh) * W + ow;
                    *reinterpret_cast<float2*>(y + pix * ldy + c) = val;
                    if (stats) {
                        psum += (double)val.x + (double)val.y;
                        psq += (double)val.x * val.x + (double)val.y * val.y;
                    }
                }
            }
        }
        if (stats) stat_add(lsum, st, n - n0, n, c, psum, psq);
    }
    if (stats) {
        __syncthreads();
        stat_flush(lsum, st, n0, N);
    }
}

// ---- the output transform in two phases through LDS (the counterpart of winograd_input_split2_kernel) ---------------------------------
// winograd_output6_kernel keeps the 8 x 8 window of its channel pair in registers and the compiler has all 64 loads (and the residual
// rows) in flight at once: 245 VGPRs, TWO waves per SIMD, and its ablation (tools/wino_variants.py) shows the load stream, the store
// stream and the arithmetic overlapping poorly (loads alone 0.165 ms, stores alone 0.159 ms, together 0.248 ms at 64x64 x 1024).
// Here a workgroup owns ONE tile x 128 consecutive channels of M (lane = channel pair: every access of a wave is 512 contiguous
// bytes) and its m + 2 waves (8 at m = 6) split the window:
//   phase A: wave j loads column j of the window (xi = (0..7, j): 8 loads per lane, issued together; M is read once: `nt`), transforms
//            it down the rows (A^T m: 8 -> 6) and leaves the 6 results in LDS [a][j][lane];
//   phase B: wave a < 6 reads row a of the intermediate, transforms it (. A: 8 -> 6), adds bias / residual, stores the 6 pixels of
//            output row a and adds its GroupNorm partial sums to the workgroup's LDS table (flushed as in the kernel above).
// ~50 VGPRs, 24 KB of LDS: four workgroups = 32 waves per CU.  Needs Cm % 128 == 0 (and Cout % 128 == 0 with the phase filters, so
// that a channel block lies inside one phase); other shapes keep the one-thread-per-window kernels.  m = 4 / m = 2 (the latent
// configurations) run the same code with 6 / 4 waves; `splits` partial sums of a split-K tile GEMM are added in order while loading.
// MO = 7 (AL = 8, round 5): the output side of F(7x7, 2x2), phase filters only (ph != 0): 7 x 7 outputs per tile from the same 8 x 8
// transform points (at_transform7); the outputs of phase (pa, pb) sit at x-grid rows 7 th + a - pa, columns 7 tw + b - pb.
template <int MO, bool RES, int TPW, int AL = MO + 2>
__global__ void __launch_bounds__(AL * 64) winograd_output_lds_kernel(const float* __restrict__ M, size_t plane, int ldm, int splits,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ res, int ldr, int res_per_image,
                                                                   float* __restrict__ y, int ldy, int N, int H, int W, int Cout,
                                                                   int cblocks, long long T, const StatArgs st, int ph, int tpw) {
    // tpw tiles (TPW = 1: one; TPW = 2: any number; consecutive, same channel block) per workgroup, the intermediate double-buffered: the waves that finish phase B of a
    // tile early -- and the two that have no output row -- already load the next tile's columns
    constexpr int NT = AL * 64, NBUF = TPW > 1 ? 2 : 1;
    constexpr int GL = 33;                                       // groups a 128-channel block can touch (cpg >= 4, unaligned start)
    constexpr int TAB = MO * 2 * 2 * GL * 2;                     // statistics table (doubles), parked in the transform buffer at the end
    constexpr int XF = NBUF * MO * AL * 64;
    constexpr bool DYN = (size_t)XF * sizeof(float2) > 65536;   // (m = 8, two buffers: 80 KB -- dynamic LDS, wino_out_lds_bytes())
    __shared__ float2 lds_static[DYN ? 1 : (XF > TAB ? XF : TAB)];
    extern __shared__ __attribute__((aligned(16))) float2 lds_dynamic[];
    float2* const lds = DYN ? lds_dynamic : lds_static;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = (int)(blockIdx.x % (unsigned)cblocks);
    const long long tile0 = (long long)(blockIdx.x / (unsigned)cblocks) * tpw;
    const int TH = wino_tdim(H, MO), TW = wino_tdim(W, MO);
    const int n0 = (int)(tile0 / ((long long)TH * TW));
    const int cm = cb * 128 + 2 * lane;                       // channel of M
    const int pq = ph ? cm / Cout : 0, c = cm - pq * Cout;    // phase filter, output channel
    const bool stats = st.s[0] != nullptr || st.s[1] != nullptr;
    // GroupNorm statistics: a thread keeps the fp64 sums of what it stores in registers -- its tpw consecutive tiles lie in at most two
    // images (slot = n - n0) -- and the workgroup reduces them ONCE, after its last tile (see below)
    double acc_s[2] = {0.0, 0.0}, acc_q[2] = {0.0, 0.0};
    const float2 b2 = bias ? *reinterpret_cast<const float2*>(bias + c) : make_float2(0.f, 0.f);
    for (int k = 0; k < tpw; ++k) {
        const long long tile = tile0 + k;
        if (tile >= T) break;
        float2* buf = lds + (k & (NBUF - 1)) * (MO * AL * 64);
        {   // ---- phase A: column `wave` of the window --------------------------------------------------------------------------
            const int j = wave;
            const float* m = M + (size_t)tile * ldm + cm + (size_t)j * plane;
            float2 v[AL], sj[MO];
#pragma unroll
            for (int i = 0; i < AL; ++i) v[i] = load_nt(reinterpret_cast<const float2*>(m + (size_t)(i * AL) * plane));
            for (int z = 1; z < splits; ++z)      // split-K tile GEMMs (small layers): partial sums M[z][xi][tiles][Cout], added in order
#pragma unroll
                for (int i = 0; i < AL; ++i)
                    v[i] = v[i] + load_nt(reinterpret_cast<const float2*>(m + ((size_t)z * (AL * AL) + i * AL) * plane));
            if constexpr (MO == 7) at_transform7(v, sj); else at_transform<MO>(v, sj);
#pragma unroll
            for (int a = 0; a < MO; ++a) buf[(a * AL + j) * 64 + lane] = sj[a];
        }
        __syncthreads();
        if (wave < MO) {   // ---- phase B: output row `wave` -------------------------------------------------------------------------
            const int a = wave;
            const int tw = (int)(tile % TW);
            const long long r = tile / TW;
            const int th = (int)(r % TH), n = (int)(r / TH);
            float2 t[AL], o[MO];
#pragma unroll
            for (int j = 0; j < AL; ++j) t[j] = buf[(a * AL + j) * 64 + lane];
            const int oh = MO == 7 ? MO * th + a - (pq >> 1) : MO * th + a;
            if (oh < H && (MO != 7 || oh >= 0)) {
                float2 rv[MO];
                if (RES) {     // the residuals of the row are fetched together (clamped addresses for the masked edge pixels)
#pragma unroll
                    for (int b = 0; b < MO; ++b) {
                        const int owc = min(MO * tw + b, W - 1);
                        const float* rp = res_per_image == 1 ? res + (size_t)n * ldr + c
                                          : res_per_image == 2 ? res + ((size_t)(n * (H >> 1) + (oh >> 1)) * (W >> 1) + (owc >> 1)) * ldr + c
                                                               : res + ((size_t)(n * H + oh) * W + owc) * ldr + c;
                        rv[b] = *reinterpret_cast<const float2*>(rp);
                    }
                }
                if constexpr (MO == 7) at_transform7(t, o); else at_transform<MO>(t, o);
                double psum = 0.0, psq = 0.0;
#pragma unroll
                for (int b = 0; b < MO; ++b) {
                    const int ow = MO == 7 ? MO * tw + b - (pq & 1) : MO * tw + b;
                    if (ow < W && (MO != 7 || ow >= 0)) {
                        float2 val = o[b] + b2;
                        if (RES) val = val + rv[b];
                        const size_t pix = ph ? (size_t)(n * 2 * H + 2 * oh + (pq >> 1)) * (2 * W) + 2 * ow + (pq & 1)
                                              : (size_t)(n * H + oh) * W + ow;
                        *reinterpret_cast<float2*>(y + pix * ldy + c) = val;
                        if (stats) {
                            psum += (double)val.x + (double)val.y;
                            psq += (double)val.x * val.x + (double)val.y * val.y;
                        }
                    }
                }
                if (stats) {
                    if (n == n0) { acc_s[0] += psum; acc_q[0] += psq; } else { acc_s[1] += psum; acc_q[1] += psq; }
                }
            }
        }
        if (NBUF == 1 && tpw > 1) __syncthreads();      // (one buffer, several tiles -- m = 8, whose two buffers would be 80 KB: the next
                                                        // tile's columns may only be written once every row of this one has been read)
    }
    if (stats) {
        // Reduction in a FIXED order (run-to-run reproducible, and free of the same-address LDS atomics that cost this kernel a third
        // of its LDS cycles in round 3): the lanes of a wave that share a group are neighbours (lane = channel pair), so a segmented
        // inclusive scan over the lanes (__shfl_up; log2(cpg / 2) steps) leaves a group's sum in its last lane; that lane parks it in
        // LDS [wave][consumer][slot][group of this channel block][sum | sq] (the transform buffer is idle by now); after a barrier one
        // thread per cell adds the MO waves' values in wave order and hands the total to the exact HBM accumulator (stats_acc.h).
        double* tab = reinterpret_cast<double*>(lds);            // [MO][2][2][GL][2] doubles <= 6 * 264 * 8 B = 12.4 KB
        static_assert(sizeof(double) == sizeof(float2), "the table is parked in the float2 transform buffer");
        const int c_first = c - 2 * lane;                        // first output channel of this 128-channel block (lane 0's)
        __syncthreads();                                         // every wave is done with the transform buffer
        if (wave < MO) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!st.s[k]) continue;
                const int g = (st.coff[k] + c) / st.cpg[k], g_first = (st.coff[k] + c_first) / st.cpg[k];
                const bool tail = sa_seg_tail(g, lane);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    double vs = acc_s[sl], vq = acc_q[sl];
                    sa_seg_scan2(vs, vq, g, lane, st.cpg[k] >> 1);      // (cpg / 2 lanes per group: lane = channel pair)
                    if (tail) {
                        double* t = tab + ((((size_t)wave * 2 + k) * 2 + sl) * GL + (g - g_first)) * 2;
                        t[0] = vs; t[1] = vq;
                    }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * 2 * GL * 2; i += NT) {
            const int which = i & 1, gl = (i >> 1) % GL, sl = ((i >> 1) / GL) & 1, k = (i >> 1) / (2 * GL);
            if (!st.s[k]) continue;
            const int n = n0 + sl;
            const int g_first = (st.coff[k] + c_first) / st.cpg[k], g_last = (st.coff[k] + c_first + 127) / st.cpg[k];
            const int g = g_first + gl;
            if (n >= N || g > g_last) continue;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < MO; ++w) v += tab[((((size_t)w * 2 + k) * 2 + sl) * GL + gl) * 2 + which];
            sa_add(st.s[k] + ((size_t)(n * 32 + g) * 2 + which) * SA_W, v);
        }
    }
}

// ---- weights: U_xi[co][ci] = (G g G^T)[i][j] in the packed 1x1 layout [xi][chunk][CoutPad][16] -----------------------
// dgrad != 0: the weights of the data-gradient convolution, g'[ci][co][r][s] = g[co][ci][2-r][2-s].
template <int MO>
__global__ void winograd_weight_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int CoutPad,
                                       int nchunks, int dgrad) {
    constexpr int AL = MO + 2;
    const int O = dgrad ? Cin : Cout, I = dgrad ? Cout : Cin;     // logical conv: O output, I input channels
    const size_t per = (size_t)nchunks * CoutPad * KC;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < per; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = idx % KC;
        size_t t = idx / KC;
        const int o = t % CoutPad;
        const int chunk = t / CoutPad;
        const int i = chunk * KC + k;
        typedef typename WinoWeightT<MO>::type WT;      // (m = 8: G g G^T in fp64, rounded to fp32 once)
        WT a[AL][3];                      // a = G g   (rows of g transformed)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            WT g[3], u[AL];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float v = 0.f;
                if (o < O && i < I)
                    v = dgrad ? w[((size_t)i * Cin + o) * 9 + (2 - r) * 3 + (2 - s)]      // w[co = i][ci = o]
                              : w[((size_t)o * Cin + i) * 9 + r * 3 + s];
                g[r] = v;
            }
            g_transform<MO>(g, u);
#pragma unroll
            for (int r = 0; r < AL; ++r) a[r][s] = u[r];
        }
#pragma unroll
        for (int r = 0; r < AL; ++r) {
            WT u[AL];
            g_transform<MO>(a[r], u);
#pragma unroll
            for (int s = 0; s < AL; ++s) p[(size_t)(r * AL + s) * per + idx] = (float)u[s];
        }
    }
}

// F(7x7, 2x2) of the four phase filters (m = 7): w4 is bbdm_upsample_phase_weights_f32's [4 Cout][Cin][3][3] tensor, whose phase
// (pa, pb) = o / Cout filter has its non-zero taps at rows pa, pa + 1 and columns pb, pb + 1: U = G g2 G^T with the 8 x 2 matrix G
// (g_transform72), same packed layout, 64 planes.
__global__ void winograd_weight72_kernel(const float* __restrict__ w4, float* __restrict__ p, int Cout4, int Cin, int CoutPad, int nchunks) {
    const int Cout = Cout4 >> 2;
    const size_t per = (size_t)nchunks * CoutPad * KC;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < per; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = idx % KC;
        size_t t = idx / KC;
        const int o = t % CoutPad;
        const int chunk = t / CoutPad;
        const int i = chunk * KC + k;
        const int pq = o < Cout4 ? o / Cout : 0, pa = pq >> 1, pb = pq & 1;
        float a[8][2];                    // a = G g2 (columns of g2 transformed)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float g[2], u[8];
#pragma unroll
            for (int r = 0; r < 2; ++r) g[r] = (o < Cout4 && i < Cin) ? w4[((size_t)o * Cin + i) * 9 + (pa + r) * 3 + (pb + s)] : 0.f;
            g_transform72(g, u);
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r][s] = u[r];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float u[8];
            g_transform72(a[r], u);
#pragma unroll
            for (int s = 0; s < 8; ++s) p[(size_t)(r * 8 + s) * per + idx] = u[s];
        }
    }
}

// ... straight into the bf16 B planes of gemm_bf3p.hip, [xi][CoutPad / 32][chunk][3][1 KB]: the same transform for a PAIR of input
// channels per thread, each U value pair split exactly and stored as bbdm_gemm_bf3p_pack_b_f32 stores it -- the same planes (up to
// the compiler's FMA contraction of G g G^T in this kernel body: <= 1 ulp of U) without the fp32 U tensor in between (4x the weights for m = 4, written and read once per optimizer step and direction: the two
// launches were 6 ms of every fourth training micro-step).
// NPL = 2 (round 6): the fp16 pair under the scale of wbound x wino_g_gain(m) -- wbound: a device float >= the filter's largest tap.
template <int MO, int NPL = 3>
__global__ void winograd_weight_planes_kernel(const float* __restrict__ w, unsigned char* __restrict__ dst, int Cout, int Cin,
                                              int CoutPad, int nchunks, int dgrad, const float* __restrict__ wbound = nullptr) {
    constexpr int AL = MO + 2;
    const int O = dgrad ? Cin : Cout, I = dgrad ? Cout : Cin;
    const size_t pairs = (size_t)nchunks * CoutPad * (KC / 2);
    const size_t xi_stride = (size_t)(CoutPad / 32) * nchunks * NPL * 1024;
    float hs = 1.f;
    if constexpr (NPL == 2) hs = h2_pow2(h2_exp_of_bound(*wbound * wino_g_gain(MO)));
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < pairs; idx += (size_t)gridDim.x * blockDim.x) {
        const int kp = (int)(idx % (KC / 2));
        size_t t = idx / (KC / 2);
        const int o = (int)(t % CoutPad);
        const int chunk = (int)(t / CoutPad);
        typedef typename WinoWeightT<MO>::type WT;
        WT a[2][AL][3];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = chunk * KC + kp * 2 + e;
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                WT g[3], u[AL];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float v = 0.f;
                    if (o < O && i < I)
                        v = dgrad ? w[((size_t)i * Cin + o) * 9 + (2 - r) * 3 + (2 - sx)] : w[((size_t)o * Cin + i) * 9 + r * 3 + sx];
                    g[r] = v;
                }
                g_transform<MO>(g, u);
#pragma unroll
                for (int r = 0; r < AL; ++r) a[e][r][sx] = u[r];
            }
        }
        unsigned char* d = dst + (((size_t)(o / 32) * nchunks + chunk) * NPL) * 1024 + ((kp * 2) >> 3) * 512 + (o & 31) * 16 +
                           ((kp * 2) & 7) * 2;
#pragma unroll
        for (int r = 0; r < AL; ++r) {
            WT u0[AL], u1[AL];
            g_transform<MO>(a[0][r], u0);
            g_transform<MO>(a[1][r], u1);
#pragma unroll
            for (int sx = 0; sx < AL; ++sx) {
                unsigned char* q = d + (size_t)(r * AL + sx) * xi_stride;
                if constexpr (NPL == 2) {
                    unsigned p1, p2;
                    h2_split2((float)u0[sx] * hs, (float)u1[sx] * hs, p1, p2);
                    *reinterpret_cast<unsigned*>(q) = p1;
                    *reinterpret_cast<unsigned*>(q + 1024) = p2;
                } else {
                    unsigned p1, p2, p3;
                    split2((float)u0[sx], (float)u1[sx], p1, p2, p3);
                    *reinterpret_cast<unsigned*>(q) = p1;
                    *reinterpret_cast<unsigned*>(q + 1024) = p2;
                    *reinterpret_cast<unsigned*>(q + 2048) = p3;
                }
            }
        }
    }
}

// conv3x3(nearest x2 (x)) as four 3x3 "phase" filters on x: output pixel (2i + a, 2j + b) reads the upsampled rows 2i + a - 1 .. 2i + a
// + 1 = x rows {i - 1, i, i} (a = 0) or {i, i, i + 1} (a = 1), so along each axis the taps collapse to [w0, w1 + w2, 0] (phase 0) or
// [0, w0 + w1, w2] (phase 1) at offsets (-1, 0, +1); zero padding of the upsampled image = zero padding of x.  w4 [4 Cout][Cin][3][3],
// filter (2a + b) Cout + co.  One thread = one (phase, co, ci).
__global__ void upsample_phase_weight_kernel(const float* __restrict__ w, float* __restrict__ w4, int Cout, int Cin) {
    const size_t n = (size_t)4 * Cout * Cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t t = i / Cin;
        const int co = (int)(t % Cout), pq = (int)(t / Cout), pa = pq >> 1, pb = pq & 1;
        const float* g = w + ((size_t)co * Cin + ci) * 9;
        float rows[3][3];                 // taps collapsed along y: rows[dy][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            rows[0][kx] = pa ? 0.f : g[kx];
            rows[1][kx] = pa ? g[kx] + g[3 + kx] : g[3 + kx] + g[6 + kx];
            rows[2][kx] = pa ? g[6 + kx] : 0.f;
        }
        float* d = w4 + i * 9;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            d[dy * 3 + 0] = pb ? 0.f : rows[dy][0];
            d[dy * 3 + 1] = pb ? rows[dy][0] + rows[dy][1] : rows[dy][1] + rows[dy][2];
            d[dy * 3 + 2] = pb ? rows[dy][2] : 0.f;
        }
    }
}

inline size_t tiles_raw(int N, int H, int W, int m) { return wino_tiles_raw(N, H, W, m); }
inline size_t tiles_padded(int N, int H, int W, int m) { return wino_tiles_padded(N, H, W, m); }
inline int planes(int m) { return wino_planes(m); }

}  // namespace

extern "C" int bbdm_upsample_phase_weights_f32(const float* w_oihw, float* w4, int Cout, int Cin, void* stream) {
    BBDM_REQUIRE(w_oihw && w4 && Cout > 0 && Cin > 0, "upsample_phase_weights: bad args");
    const size_t n = (size_t)4 * Cout * Cin;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(upsample_phase_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, w4, Cout, Cin);
    BBDM_CHECK_LAUNCH("upsample_phase_weights");
    return BBDM_OK;
}

extern "C" size_t bbdm_winograd_packed_floats(int m, int Cout, int CinPad) {
    return (size_t)planes(m) * cdiv(CinPad, KC) * (cdiv(Cout, 128) * 128) * KC;
}

extern "C" int bbdm_winograd_pack_weight_f32(int m, const float* w_oihw, float* packed, int Cout, int Cin, int InPad,
                                             int dgrad, void* stream) {
    // forward: conv Cin -> Cout, input tensor carries InPad >= Cin channels.
    // dgrad  : conv Cout -> Cin, its input (dY) carries InPad >= Cout channels.
    // m = 7  : w_oihw = the [Cout = 4 C][Cin][3][3] phase filters bbdm_upsample_phase_weights_f32 wrote; packed = G g2 G^T of F(7x7, 2x2)
    BBDM_WINO_M78(m);
    BBDM_REQUIRE(w_oihw && packed && Cout > 0 && Cin > 0 && InPad % 4 == 0, "winograd_pack: bad args");
    BBDM_REQUIRE(InPad >= (dgrad ? Cout : Cin), "winograd_pack: InPad too small");
    BBDM_REQUIRE(m != 7 || (!dgrad && Cout % 4 == 0), "winograd_pack: m = 7 takes the 4 C phase filters of a forward conv");
    const int O = dgrad ? Cin : Cout;
    const int CoutPad = cdiv(O, 128) * 128, nchunks = cdiv(InPad, KC);
    const size_t per = (size_t)nchunks * CoutPad * KC;
    int blocks = (int)((per + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (m == 7)
        hipLaunchKernelGGL(winograd_weight72_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout, Cin,
                           CoutPad, nchunks);
    else if (m == 2)
        hipLaunchKernelGGL(winograd_weight_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout,
                           Cin, CoutPad, nchunks, dgrad);
    else if (m == 4)
        hipLaunchKernelGGL(winograd_weight_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout,
                           Cin, CoutPad, nchunks, dgrad);
    else if (m == 8)
        hipLaunchKernelGGL(winograd_weight_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout,
                           Cin, CoutPad, nchunks, dgrad);
    else
        hipLaunchKernelGGL(winograd_weight_kernel<6>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout,
                           Cin, CoutPad, nchunks, dgrad);
    BBDM_CHECK_LAUNCH("winograd_pack");
    return BBDM_OK;
}

// ... with the B planes of gemm_bf3p.hip as the destination (= bbdm_winograd_pack_weight_f32 + bbdm_gemm_bf3p_pack_b_f32 up to FMA contraction;
// b_planes: bbdm_gemm_bf3p_b_bytes((m + 2)^2, InPad, O) bytes, InPad % 16 == 0)
static int winograd_pack_planes(int m, const float* w_oihw, void* b_planes, int Cout, int Cin, int InPad, int dgrad, const float* wbound,
                                void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(w_oihw && b_planes && Cout > 0 && Cin > 0 && InPad % KC == 0, "winograd_pack_bf3p: bad args (InPad %% 16)");
    BBDM_REQUIRE(InPad >= (dgrad ? Cout : Cin), "winograd_pack_bf3p: InPad too small");
    BBDM_REQUIRE(((uintptr_t)b_planes & 15) == 0, "winograd_pack_bf3p: b_planes alignment");
    const int O = dgrad ? Cin : Cout;
    const int CoutPad = cdiv(O, 128) * 128, nchunks = InPad / KC;
    const size_t pairs = (size_t)nchunks * CoutPad * (KC / 2);
    int blocks = (int)((pairs + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    unsigned char* d = (unsigned char*)b_planes;
#define BBDM_WINO_PK(MO)                                                                                                          \
    do {                                                                                                                          \
        if (wbound)                                                                                                               \
            hipLaunchKernelGGL((winograd_weight_planes_kernel<MO, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, d, Cout, \
                               Cin, CoutPad, nchunks, dgrad, wbound);                                                             \
        else                                                                                                                      \
            hipLaunchKernelGGL((winograd_weight_planes_kernel<MO, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, d, Cout, \
                               Cin, CoutPad, nchunks, dgrad, nullptr);                                                            \
    } while (0)
    if (m == 2) BBDM_WINO_PK(2); else if (m == 4) BBDM_WINO_PK(4); else if (m == 8) BBDM_WINO_PK(8); else BBDM_WINO_PK(6);
#undef BBDM_WINO_PK
    BBDM_CHECK_LAUNCH("winograd_pack_planes");
    return BBDM_OK;
}
extern "C" int bbdm_winograd_pack_weight_bf3p_f32(int m, const float* w_oihw, void* b_planes, int Cout, int Cin, int InPad, int dgrad,
                                                  void* stream) {
    return winograd_pack_planes(m, w_oihw, b_planes, Cout, Cin, InPad, dgrad, nullptr, stream);
}
// ... as the fp16-pair planes (bbdm_gemm_h2p_b_bytes bytes) under the scale of wbound x bbdm_winograd_g_gain(m); wbound: a device float
// >= the filter's largest tap (bbdm_absmax_f32 of the weight tensor), the same pointer bbdm_winograd_gemm_h2p_f32 takes
extern "C" int bbdm_winograd_pack_weight_h2p_f32(int m, const float* w_oihw, void* b_planes, int Cout, int Cin, int InPad, int dgrad,
                                                 const float* wbound, void* stream) {
    BBDM_REQUIRE(wbound, "winograd_pack_h2p: null bound");
    return winograd_pack_planes(m, w_oihw, b_planes, Cout, Cin, InPad, dgrad, wbound, stream);
}
extern "C" float bbdm_winograd_g_gain(int m) { return wino_g_gain(m); }

extern "C" size_t bbdm_winograd_tiles(int m, int N, int H, int W) {
    return (m == 2 || m == 4 || m == 6 || m == 7 || m == 8) ? tiles_padded(N, H, W, m) : 0;
}

extern "C" size_t bbdm_winograd_workspace_floats(int m, int N, int H, int W, int CinPad, int Cout) {
    if (m != 2 && m != 4 && m != 6 && m != 8) return 0;
    return (size_t)planes(m) * tiles_padded(N, H, W, m) * ((size_t)CinPad + (size_t)Cout);
}

static int winograd_input_planes(int m, const float* x, int ldx, void* Vp, void* Vt, const float* pre_scale, const float* pre_bias,
                                 int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad, void* stream,
                                 const GnFold* fold, bool f32out, const float* hbound = nullptr, int vt_h2 = 0);

extern "C" int bbdm_winograd_input_f32(int m, const float* x, int ldx, float* V, const float* pre_scale,
                                       const float* pre_bias, int pre_ld, int pre_silu, int upsample, int N, int H, int W,
                                       int CinPad, void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(m != 8 || CinPad % 32 == 0, "winograd_input: m = 8 takes whole 32-channel chunks");
    BBDM_REQUIRE(x && V && N > 0, "winograd_input: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    BBDM_REQUIRE(!upsample || (H % 2 == 0 && W % 2 == 0), "winograd_input: upsample needs even H, W");
    BBDM_REQUIRE(CinPad > 0 && CinPad % 4 == 0 && ldx % 4 == 0 && ldx >= CinPad, "winograd_input: CinPad=%d ldx=%d", CinPad, ldx);
    BBDM_REQUIRE((((uintptr_t)x | (uintptr_t)V) & 15) == 0, "winograd_input: 16-byte alignment");
    BBDM_REQUIRE((pre_scale == nullptr) == (pre_bias == nullptr), "winograd_input: pre_scale / pre_bias must come together");
    BBDM_REQUIRE(!pre_scale || (pre_ld % 4 == 0 && pre_ld >= CinPad && (((uintptr_t)pre_scale | (uintptr_t)pre_bias) & 15) == 0),
                 "winograd_input: pre_ld / alignment of the fused-producer coefficients");
    const size_t T = tiles_raw(N, H, W, m), Tp = tiles_padded(N, H, W, m);
    // whole 32-channel chunks: the two-phase LDS kernel, storing fp32 rows (round 5; rows of padding tiles are written as zeros); other
    // channel counts keep the one-thread-per-window kernels below
    if (CinPad % 32 == 0 && Tp < (1ull << 31))
        return winograd_input_planes(m, x, ldx, V, nullptr, pre_scale, pre_bias, pre_ld, pre_silu, upsample, N, H, W, CinPad, stream,
                                     nullptr, true);
    const size_t vplane = Tp * (size_t)CinPad;
    const long long units = (long long)T * (CinPad / (m == 6 ? 2 : 4));
    long long blocks = (units + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const dim3 g((unsigned)blocks), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (m == 6) {
#define BBDM_WINO_IN6(PRE, UP)                                                                                            \
    hipLaunchKernelGGL((winograd_input6_kernel<PRE, UP>), g, b, 0, st, x, ldx, V, pre_scale, pre_bias, pre_ld, pre_silu, N, \
                       H, W, CinPad, vplane)
        if (pre_scale) { if (upsample) BBDM_WINO_IN6(true, true); else BBDM_WINO_IN6(true, false); }
        else           { if (upsample) BBDM_WINO_IN6(false, true); else BBDM_WINO_IN6(false, false); }
#undef BBDM_WINO_IN6
        BBDM_CHECK_LAUNCH("winograd_input");
        return BBDM_OK;
    }
#define BBDM_WINO_IN(MO, PRE, UP)                                                                                         \
    hipLaunchKernelGGL((winograd_input_kernel<MO, PRE, UP>), g, b, 0, st, x, ldx, V, pre_scale, pre_bias, pre_ld, pre_silu, \
                       N, H, W, CinPad, vplane)
#define BBDM_WINO_IN_M(MO)                                                                      \
    do {                                                                                        \
        if (pre_scale) { if (upsample) BBDM_WINO_IN(MO, true, true); else BBDM_WINO_IN(MO, true, false); }   \
        else           { if (upsample) BBDM_WINO_IN(MO, false, true); else BBDM_WINO_IN(MO, false, false); } \
    } while (0)
    if (m == 2) BBDM_WINO_IN_M(2); else BBDM_WINO_IN_M(4);
#undef BBDM_WINO_IN_M
#undef BBDM_WINO_IN
    BBDM_CHECK_LAUNCH("winograd_input");
    return BBDM_OK;
}

extern "C" int bbdm_winograd_gemm_f32(int m, const float* V, const float* packed_wino, float* M, int N, int H, int W,
                                      int CinPad, int Cout, void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(V && packed_wino && M && N > 0, "winograd_gemm: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    BBDM_REQUIRE(CinPad > 0 && CinPad % 4 == 0 && Cout % 4 == 0 && Cout > 0, "winograd_gemm: bad channel counts");
    BBDM_REQUIRE((((uintptr_t)V | (uintptr_t)M | (uintptr_t)packed_wino) & 15) == 0, "winograd_gemm: 16-byte alignment");
    const size_t Tp = tiles_padded(N, H, W, m);
    BBDM_REQUIRE(Tp * (size_t)CinPad < (1ull << 32), "winograd_gemm: one transformed plane exceeds 2^32 elements");
    const size_t wz = bbdm_winograd_packed_floats(m, Cout, CinPad) / planes(m);
    return bbdm_conv1x1_batched(V, CinPad, Tp * (size_t)CinPad, packed_wino, wz, M, Cout, Tp * (size_t)Cout, planes(m),
                                (int)(Tp / 32), 32, CinPad, Cout, (hipStream_t)stream);
}

// The same (m+2)^2 GEMMs on the BF16 matrix core with fp32 accuracy (gemm_bf3.hip): packed_bf3 = the output of
// bbdm_gemm_bf3_pack_f32 applied to the buffer bbdm_winograd_pack_weight_f32 filled (batch = (m+2)^2).
extern "C" int bbdm_winograd_gemm_bf3_f32(int m, const float* V, const void* packed_bf3, float* M, int N, int H, int W,
                                          int CinPad, int Cout, void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(V && packed_bf3 && M && N > 0, "winograd_gemm_bf3: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    return bbdm_gemm_bf3_f32(V, packed_bf3, M, planes(m), (long long)tiles_padded(N, H, W, m), CinPad, Cout, stream);
}

// The input transform that writes the three bf16 planes of gemm_bf3p.hip (same arguments as bbdm_winograd_input_f32; Vp holds
// bbdm_gemm_bf3p_a_bytes((m+2)^2, tiles, CinPad) bytes; CinPad a multiple of 16), and the tile GEMMs on them: b_planes =
// bbdm_gemm_bf3p_pack_b_f32 applied to the buffer bbdm_winograd_pack_weight_f32 filled (batch = (m+2)^2).
static int winograd_input_planes(int m, const float* x, int ldx, void* Vp, void* Vt, const float* pre_scale, const float* pre_bias,
                                 int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad, void* stream,
                                 const GnFold* fold, bool f32out, const float* hbound, int vt_h2) {
    BBDM_WINO_M78(m);
    BBDM_REQUIRE(!hbound || !f32out, "winograd_input_h2p: the fp16-pair planes have no fp32 form");
    BBDM_REQUIRE(m != 7 || (!upsample && !Vt && !fold && !f32out), "winograd_input_bf3p: m = 7 (phase filters) takes x itself, planes only");
    BBDM_REQUIRE(m != 8 || !fold, "winograd_input_bf3p: m = 8 is a tile of the large layers (no coefficient folding)");
    BBDM_REQUIRE(x && Vp && N > 0, "winograd_input_bf3p: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    BBDM_REQUIRE(!upsample || (H % 2 == 0 && W % 2 == 0), "winograd_input_bf3p: upsample needs even H, W");
    BBDM_REQUIRE(CinPad > 0 && CinPad % KC == 0 && ldx % 2 == 0 && ldx >= CinPad, "winograd_input_bf3p: CinPad=%d ldx=%d", CinPad, ldx);
    BBDM_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)Vp & 15) == 0, "winograd_input_bf3p: alignment");
    BBDM_REQUIRE((pre_scale == nullptr) == (pre_bias == nullptr), "winograd_input_bf3p: pre_scale / pre_bias must come together");
    BBDM_REQUIRE(!pre_scale || (pre_ld % 2 == 0 && pre_ld >= CinPad && (((uintptr_t)pre_scale | (uintptr_t)pre_bias) & 7) == 0),
                 "winograd_input_bf3p: pre_ld / alignment of the fused-producer coefficients");
    const size_t T = tiles_raw(N, H, W, m), Tp = tiles_padded(N, H, W, m);
    const int nchunks = CinPad / KC, TG = (int)(Tp / 8);
    const size_t plane = Tp * (size_t)CinPad * (f32out || hbound ? 4 : 6);  // bytes of one transform point
    const size_t plane_t = (size_t)((CinPad + 31) / 32 * 32) * Tp * (vt_h2 ? 4 : 6);       // ... of the transposed copy (whole 32-channel row groups)
    BBDM_REQUIRE(!Vt || (!upsample && ((uintptr_t)Vt & 15) == 0), "winograd_input_bf3p: the transposed copy needs upsample = 0, 16-B alignment");
    hipStream_t st = (hipStream_t)stream;
    {
        const long long blocks = 8ll * ((TG + 7) / 8) * nchunks;
        BBDM_REQUIRE(blocks < (1ll << 31) && Tp < (1ull << 31), "winograd_input_bf3p: too many workgroups / tiles");
        const dim3 g((unsigned)blocks);
        const int Hs = upsample ? H / 2 : H, Ws = upsample ? W / 2 : W;
        // 32-bit element indices + 24-bit row multiplies where the tensor allows it (every shape of the reference's templates)
        // (option wino_idx64 = 1 forces the 64-bit variant: the tests run both on ordinary shapes)
        const int idx64_env = bbdm_option(BBDM_OPT_WINO_IDX64);
        const bool idx64 = idx64_env || (unsigned long long)N * Hs * Ws * (unsigned long long)ldx >= (1ull << 32) ||
                           (unsigned long long)Ws * (unsigned long long)ldx >= (1ull << 24) || Hs >= (1 << 24);
        const FastDiv dTW = fastdiv_make((unsigned)wino_tdim(W, m)), dTH = fastdiv_make((unsigned)wino_tdim(H, m)),
                      dCH = fastdiv_make((unsigned)nchunks);
        const GnFold gn = fold ? *fold : GnFold{};
        if (m == 7) {                     // F(7x7, 2x2): the m = 6 kernel with windows 7 pixels apart
#define BBDM_WINO_INS2_7(PRE, I64)                                                                                                \
    hipLaunchKernelGGL((winograd_input_split2_kernel<6, PRE, false, false, I64, false, false, 7>), g, dim3(8 * 64), 0, st, x, ldx,   \
                       (unsigned char*)Vp, pre_scale, pre_bias, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane, nullptr,  \
                       plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2)
#define BBDM_WINO_INS2_7H(PRE, I64)                                                                                               \
    hipLaunchKernelGGL((winograd_input_split2_kernel<6, PRE, false, false, I64, false, false, 7, 2>), g, dim3(8 * 64), 0, st, x, ldx, \
                       (unsigned char*)Vp, pre_scale, pre_bias, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane, nullptr,  \
                       plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2)
            if (hbound) {
                if (pre_scale) { if (idx64) BBDM_WINO_INS2_7H(true, true); else BBDM_WINO_INS2_7H(true, false); }
                else           { if (idx64) BBDM_WINO_INS2_7H(false, true); else BBDM_WINO_INS2_7H(false, false); }
            }
            else if (pre_scale) { if (idx64) BBDM_WINO_INS2_7(true, true); else BBDM_WINO_INS2_7(true, false); }
            else           { if (idx64) BBDM_WINO_INS2_7(false, true); else BBDM_WINO_INS2_7(false, false); }
#undef BBDM_WINO_INS2_7H
#undef BBDM_WINO_INS2_7
            BBDM_CHECK_LAUNCH("winograd_input_bf3p(m = 7)");
            return BBDM_OK;
        }
        if (fold) {                       // the kernel forms the coefficients (small problems; 32-bit indices, no transposed copy)
            BBDM_REQUIRE(!idx64 && !Vt, "winograd_input_bf3p_gn: tensor too large for the coefficient-folding variant");
#define BBDM_WINO_INS2_G(MO)                                                                                                      \
    do {                                                                                                                          \
        if (upsample)                                                                                                             \
            hipLaunchKernelGGL((winograd_input_split2_kernel<MO, true, true, false, false, true>), g, dim3((MO + 2) * 64), 0, st, x, ldx, \
                               (unsigned char*)Vp, nullptr, nullptr, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane,  \
                               nullptr, plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2);                                              \
        else                                                                                                                      \
            hipLaunchKernelGGL((winograd_input_split2_kernel<MO, true, false, false, false, true>), g, dim3((MO + 2) * 64), 0, st, x, ldx, \
                               (unsigned char*)Vp, nullptr, nullptr, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane,  \
                               nullptr, plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2);                                              \
    } while (0)
#define BBDM_WINO_INS2_GH(MO)                                                                                                     \
    do {                                                                                                                          \
        if (upsample)                                                                                                             \
            hipLaunchKernelGGL((winograd_input_split2_kernel<MO, true, true, false, false, true, false, MO, 2>), g, dim3((MO + 2) * 64), 0, st, x, ldx, \
                               (unsigned char*)Vp, nullptr, nullptr, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane,  \
                               nullptr, plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2);                                      \
        else                                                                                                                      \
            hipLaunchKernelGGL((winograd_input_split2_kernel<MO, true, false, false, false, true, false, MO, 2>), g, dim3((MO + 2) * 64), 0, st, x, ldx, \
                               (unsigned char*)Vp, nullptr, nullptr, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane,  \
                               nullptr, plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2);                                      \
    } while (0)
            if (hbound) { if (m == 2) BBDM_WINO_INS2_GH(2); else if (m == 4) BBDM_WINO_INS2_GH(4); else BBDM_WINO_INS2_GH(6); }
            else if (m == 2) BBDM_WINO_INS2_G(2); else if (m == 4) BBDM_WINO_INS2_G(4); else BBDM_WINO_INS2_G(6);
#undef BBDM_WINO_INS2_GH
#undef BBDM_WINO_INS2_G
            BBDM_CHECK_LAUNCH("winograd_input_bf3p_gn");
            return BBDM_OK;
        }
        if (f32out) {                     // fp32 rows instead of planes (no transposed copy, no coefficient folding)
            BBDM_REQUIRE(!Vt && CinPad % 32 == 0, "winograd_input: the fp32 form has no transposed copy and takes whole 32-channel chunks");
            // its workgroups own 4 tiles x 32 channels: re-derive the grid quantities the kernel reads
            const int nchunks = CinPad / 32, TG = (int)(Tp / 4);
            const FastDiv dCH = fastdiv_make((unsigned)nchunks);
            const dim3 g((unsigned)(8ll * ((TG + 7) / 8) * nchunks));
            BBDM_REQUIRE(8ll * ((TG + 7) / 8) * nchunks < (1ll << 31), "winograd_input: too many workgroups");
#define BBDM_WINO_INS2_F(MO, PRE, UP, I64)                                                                                        \
    hipLaunchKernelGGL((winograd_input_split2_kernel<MO, PRE, UP, false, I64, false, true>), g, dim3((MO + 2) * 64), 0, st, x, ldx, \
                       (unsigned char*)Vp, pre_scale, pre_bias, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane, nullptr, \
                       plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2)
#define BBDM_WINO_INS2_FI(MO, PRE, UP) do { if (idx64) BBDM_WINO_INS2_F(MO, PRE, UP, true); else BBDM_WINO_INS2_F(MO, PRE, UP, false); } while (0)
#define BBDM_WINO_INS2_FM(MO)                                                                                                     \
    do {                                                                                                                          \
        if (pre_scale) { if (upsample) BBDM_WINO_INS2_FI(MO, true, true); else BBDM_WINO_INS2_FI(MO, true, false); }              \
        else           { if (upsample) BBDM_WINO_INS2_FI(MO, false, true); else BBDM_WINO_INS2_FI(MO, false, false); }            \
    } while (0)
            if (m == 2) BBDM_WINO_INS2_FM(2); else if (m == 4) BBDM_WINO_INS2_FM(4); else if (m == 8) BBDM_WINO_INS2_FM(8); else BBDM_WINO_INS2_FM(6);
#undef BBDM_WINO_INS2_FM
#undef BBDM_WINO_INS2_FI
#undef BBDM_WINO_INS2_F
            BBDM_CHECK_LAUNCH("winograd_input(fp32 rows, two-phase)");
            return BBDM_OK;
        }
#define BBDM_WINO_INS2_I(MO, PRE, UP, TR, I64)                                                                                    \
    hipLaunchKernelGGL((winograd_input_split2_kernel<MO, PRE, UP, TR, I64>), g, dim3((MO + 2) * 64), 0, st, x, ldx, (unsigned char*)Vp, \
                       pre_scale, pre_bias, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane, (unsigned char*)Vt,   \
                       plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2)
#define BBDM_WINO_INS2(MO, PRE, UP, TR) do { if (idx64) BBDM_WINO_INS2_I(MO, PRE, UP, TR, true); else BBDM_WINO_INS2_I(MO, PRE, UP, TR, false); } while (0)
#define BBDM_WINO_INS2_M(MO)                                                                        \
    do {                                                                                            \
        if (Vt) { if (pre_scale) BBDM_WINO_INS2(MO, true, false, true); else BBDM_WINO_INS2(MO, false, false, true); }   \
        else if (pre_scale) { if (upsample) BBDM_WINO_INS2(MO, true, true, false); else BBDM_WINO_INS2(MO, true, false, false); }   \
        else           { if (upsample) BBDM_WINO_INS2(MO, false, true, false); else BBDM_WINO_INS2(MO, false, false, false); } \
    } while (0)
#define BBDM_WINO_INS2_HI(MO, PRE, UP, TR, I64)                                                                                   \
    hipLaunchKernelGGL((winograd_input_split2_kernel<MO, PRE, UP, TR, I64, false, false, MO, 2>), g, dim3((MO + 2) * 64), 0, st, x, ldx, \
                       (unsigned char*)Vp, pre_scale, pre_bias, pre_ld, pre_silu, N, H, W, nchunks, (unsigned)T, TG, plane,           \
                       (unsigned char*)Vt, plane_t, (int)(Tp / 16), dTW, dTH, dCH, gn, hbound, vt_h2)
#define BBDM_WINO_INS2_H(MO, PRE, UP, TR) do { if (idx64) BBDM_WINO_INS2_HI(MO, PRE, UP, TR, true); else BBDM_WINO_INS2_HI(MO, PRE, UP, TR, false); } while (0)
#define BBDM_WINO_INS2_HM(MO)                                                                       \
    do {                                                                                            \
        if (Vt) { if (pre_scale) BBDM_WINO_INS2_H(MO, true, false, true); else BBDM_WINO_INS2_H(MO, false, false, true); }   \
        else if (pre_scale) { if (upsample) BBDM_WINO_INS2_H(MO, true, true, false); else BBDM_WINO_INS2_H(MO, true, false, false); }   \
        else           { if (upsample) BBDM_WINO_INS2_H(MO, false, true, false); else BBDM_WINO_INS2_H(MO, false, false, false); } \
    } while (0)
        if (hbound) { if (m == 2) BBDM_WINO_INS2_HM(2); else if (m == 4) BBDM_WINO_INS2_HM(4); else if (m == 8) BBDM_WINO_INS2_HM(8); else BBDM_WINO_INS2_HM(6); }
        else if (m == 2) BBDM_WINO_INS2_M(2); else if (m == 4) BBDM_WINO_INS2_M(4); else if (m == 8) BBDM_WINO_INS2_M(8); else BBDM_WINO_INS2_M(6);
#undef BBDM_WINO_INS2_HM
#undef BBDM_WINO_INS2_H
#undef BBDM_WINO_INS2_HI
#undef BBDM_WINO_INS2_M
#undef BBDM_WINO_INS2
#undef BBDM_WINO_INS2_I
        BBDM_CHECK_LAUNCH("winograd_input_bf3p");
        return BBDM_OK;
    }
}

extern "C" int bbdm_winograd_input_bf3p_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale,
                                            const float* pre_bias, int pre_ld, int pre_silu, int upsample, int N, int H, int W,
                                            int CinPad, void* stream) {
    return winograd_input_planes(m, x, ldx, Vp, nullptr, pre_scale, pre_bias, pre_ld, pre_silu, upsample, N, H, W, CinPad, stream, nullptr, false);
}

// The same transform with GroupNorm [-> FiLM] [-> SiLU] folded in FROM THE STATISTICS (util.py:214-216 + openaimodel.py:258-278): the
// kernel forms x * sc[n][c] + bi[n][c] from stats (the accumulator slot of this GroupNorm, [N][G][2] values in the layout of
// bbdm_groupnorm_stats_bytes), gamma / beta [C] and the FiLM vector film[n][c] (scale), film[n][C + c] (shift) (null: none) -- the
// expressions and therefore the bits of bbdm_groupnorm_coeffs_f32 followed by bbdm_winograd_input_bf3p_f32, one launch instead of
// two.  For SMALL problems (every thread repeats the fp64 fold for its channel pair).  Argument order of bbdm_winograd_input_bf3p_f32
// with (stats, unused) in place of (pre_scale, pre_bias) and C in place of pre_ld; the GroupNorm's parameters follow CinPad.
// C == CinPad, C / G even, HW = pixels per image of the NORMALISED tensor (x's own H * W, also when upsample = 1).
extern "C" int bbdm_winograd_input_bf3p_gn_f32(int m, const float* x, int ldx, void* Vp, const void* stats, const void* unused, int C,
                                               int pre_silu, int upsample, int N, int H, int W, int CinPad, const float* gamma,
                                               const float* beta, const float* film, int film_ld, int HW, int G, float eps,
                                               void* stream) {
    (void)unused;
    BBDM_REQUIRE(stats && gamma && beta, "winograd_input_bf3p_gn: null pointer");
    BBDM_REQUIRE(C == CinPad && G > 0 && C % G == 0 && (C / G) % 2 == 0 && HW > 0 && (!film || film_ld >= 2 * C),
                 "winograd_input_bf3p_gn: C=%d CinPad=%d G=%d HW=%d film_ld=%d", C, CinPad, G, HW, film_ld);
    BBDM_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)film) & 7) == 0 && (!film || film_ld % 2 == 0),
                 "winograd_input_bf3p_gn: gamma / beta / film must be 8-byte aligned");
    GnFold f;
    f.stats = (const unsigned long long*)stats; f.gamma = gamma; f.beta = beta; f.film = film;
    f.film_ld = film_ld; f.C = C; f.G = G; f.cpg = C / G; f.cnt = (double)HW * (double)(C / G); f.eps = eps;
    return winograd_input_planes(m, x, ldx, Vp, nullptr, nullptr, nullptr, C, pre_silu, upsample, N, H, W, CinPad, stream, &f, false);
}

// ... and, for the training forward, ALSO the transposed planes Vt (bbdm_gemm_bf3p_tn_at_bytes((m+2)^2, tiles, CinPad) bytes): the A
// operand of the Winograd-domain weight gradient dU_xi = V_xi^T dM_xi on the same bf16x3 GEMM (bbdm_gemm_bf3p_tn_f32).
// (same argument order as bbdm_winograd_input_bf3p_f32, Vt appended; upsample must be 0)
extern "C" int bbdm_winograd_input_bf3p_tr_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale,
                                               const float* pre_bias, int pre_ld, int pre_silu, int upsample, int N, int H, int W,
                                               int CinPad, void* Vt, void* stream) {
    BBDM_REQUIRE(Vt && !upsample, "winograd_input_bf3p_tr: null pointer / upsample != 0");
    return winograd_input_planes(m, x, ldx, Vp, Vt, pre_scale, pre_bias, pre_ld, pre_silu, 0, N, H, W, CinPad, stream, nullptr, false);
}

// ---- the same stages on the fp16-pair planes (round 6; h2_split.h, gemm_bf3p.hip "h2"): Vp holds bbdm_gemm_h2p_a_bytes((m+2)^2, tiles,
// CinPad) bytes, b_planes = bbdm_gemm_h2p_pack_b_f32 of the buffer bbdm_winograd_pack_weight_f32 filled; vbound: a device float >=
// max |d| of the TRANSFORMED tensor (x after the fused producer; both stages multiply it by bbdm_winograd_input_gain(m) themselves),
// ubound: >= the largest TAP of the filter (bbdm_absmax_f32 of the weights; packer and GEMM multiply it by bbdm_winograd_g_gain(m)).
extern "C" float bbdm_winograd_input_gain(int m) { return wino_input_gain(m); }
extern "C" int bbdm_winograd_input_h2p_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias,
                                           int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad,
                                           const float* vbound, void* stream) {
    BBDM_REQUIRE(vbound, "winograd_input_h2p: null bound");
    return winograd_input_planes(m, x, ldx, Vp, nullptr, pre_scale, pre_bias, pre_ld, pre_silu, upsample, N, H, W, CinPad, stream, nullptr,
                                 false, vbound);
}
// ... and, for the training forward, ALSO the transposed planes Vt of the weight gradient -- those stay the exact bf16 split
// (bbdm_gemm_bf3p_tn_at_bytes bytes, the layout of bbdm_winograd_input_bf3p_tr_f32: dM, their partner in that GEMM, has no bounded range)
extern "C" int bbdm_winograd_input_h2p_tr_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias,
                                              int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad, void* Vt,
                                              const float* vbound, void* stream) {
    BBDM_REQUIRE(Vt && vbound && !upsample, "winograd_input_h2p_tr: null pointer / upsample != 0");
    return winograd_input_planes(m, x, ldx, Vp, Vt, pre_scale, pre_bias, pre_ld, pre_silu, 0, N, H, W, CinPad, stream, nullptr, false, vbound);
}
// ... with the transposed copy on the fp16 pair as well (bbdm_gemm_h2p_tn_at_bytes bytes; the weight gradient then runs bbdm_gemm_h2p_tn_f32)
extern "C" int bbdm_winograd_input_h2p_tr2_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias,
                                               int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad, void* Vt,
                                               const float* vbound, void* stream) {
    BBDM_REQUIRE(Vt && vbound && !upsample, "winograd_input_h2p_tr2: null pointer / upsample != 0");
    return winograd_input_planes(m, x, ldx, Vp, Vt, pre_scale, pre_bias, pre_ld, pre_silu, 0, N, H, W, CinPad, stream, nullptr, false, vbound, 1);
}
extern "C" int bbdm_winograd_input_h2p_gn_f32(int m, const float* x, int ldx, void* Vp, const void* stats, const void* unused, int C,
                                              int pre_silu, int upsample, int N, int H, int W, int CinPad, const float* gamma,
                                              const float* beta, const float* film, int film_ld, int HW, int G, float eps,
                                              const float* vbound, void* stream) {
    (void)unused;
    BBDM_REQUIRE(stats && gamma && beta && vbound, "winograd_input_h2p_gn: null pointer");
    BBDM_REQUIRE(C == CinPad && G > 0 && C % G == 0 && (C / G) % 2 == 0 && HW > 0 && (!film || film_ld >= 2 * C),
                 "winograd_input_h2p_gn: C=%d CinPad=%d G=%d HW=%d film_ld=%d", C, CinPad, G, HW, film_ld);
    BBDM_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)film) & 7) == 0 && (!film || film_ld % 2 == 0),
                 "winograd_input_h2p_gn: gamma / beta / film must be 8-byte aligned");
    GnFold f;
    f.stats = (const unsigned long long*)stats; f.gamma = gamma; f.beta = beta; f.film = film;
    f.film_ld = film_ld; f.C = C; f.G = G; f.cpg = C / G; f.cnt = (double)HW * (double)(C / G); f.eps = eps;
    return winograd_input_planes(m, x, ldx, Vp, nullptr, nullptr, nullptr, C, pre_silu, upsample, N, H, W, CinPad, stream, &f, false, vbound);
}
extern "C" int bbdm_winograd_gemm_h2p_splitk_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W, int CinPad,
                                                 int Cout, int splits, const float* vbound, const float* ubound, void* stream) {
    BBDM_WINO_M78(m);
    BBDM_REQUIRE(Vp && b_planes && M && vbound && ubound && N > 0, "winograd_gemm_h2p: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    return bbdm_gemm_h2p_gain_splitk(Vp, b_planes, vbound, wino_input_gain(m), ubound, wino_g_gain(m), M, Cout, planes(m),
                                     (long long)tiles_padded(N, H, W, m), (long long)((tiles_raw(N, H, W, m) + 31) / 32 * 32), CinPad, Cout,
                                     splits, stream);
}
extern "C" int bbdm_winograd_gemm_h2p_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W, int CinPad,
                                          int Cout, const float* vbound, const float* ubound, void* stream) {
    return bbdm_winograd_gemm_h2p_splitk_f32(m, Vp, b_planes, M, N, H, W, CinPad, Cout, 1, vbound, ubound, stream);
}

// splits > 1 (small layers, bbdm_winograd_gemm_bf3p_splits): the partial sums of split z go to M[z][(m+2)^2][tiles][Cout] and
// bbdm_winograd_output_splitk_stats_f32 adds them.  Only the row tiles holding real tiles are computed (tiles_raw rounded up to 32
// rows: an 8x8 latent at batch 32 fills 128 of its 256 padded rows).
extern "C" int bbdm_winograd_gemm_bf3p_splits(int m, int N, int H, int W, int CinPad, int Cout) {
    if ((m != 2 && m != 4) || N <= 0 || H <= 0 || W <= 0) return 1;      // (m = 6 is chosen for layers that fill the chip many times over)
    const int s = bbdm_gemm_bf3p_fwd_splits(planes(m), (long long)((tiles_raw(N, H, W, m) + 31) / 32 * 32), CinPad, Cout);
    return s > 1 ? s : 1;
}
extern "C" int bbdm_winograd_gemm_bf3p_splitk_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W,
                                                  int CinPad, int Cout, int splits, void* stream) {
    BBDM_WINO_M78(m);
    BBDM_REQUIRE(Vp && b_planes && M && N > 0, "winograd_gemm_bf3p: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    return bbdm_gemm_bf3p_splitk_f32(Vp, b_planes, M, Cout, planes(m), (long long)tiles_padded(N, H, W, m),
                                     (long long)((tiles_raw(N, H, W, m) + 31) / 32 * 32), CinPad, Cout, splits, stream);
}
extern "C" int bbdm_winograd_gemm_bf3p_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W,
                                           int CinPad, int Cout, void* stream) {
    return bbdm_winograd_gemm_bf3p_splitk_f32(m, Vp, b_planes, M, N, H, W, CinPad, Cout, 1, stream);
}

// stats0 / stats1 (each may be NULL): the GroupNorm accumulators (sum, sum of squares per image and group) of up to two consumers
// of `out`, each bbdm_groupnorm_stats_bytes(N, 32) bytes = [N][32][2][SA_W] 64-bit limb words (stats_acc.h -- NOT [N][32][2] doubles:
// a caller sizing them as doubles would under-allocate 4x); cpg = channels per group of that consumer (a multiple of 4), coff = channel
// offset of `out` in the consumer's tensor.  The caller zeroes them; the kernel ADDS (several producers may fill one consumer's
// statistics).
extern "C" int bbdm_winograd_output_splitk_stats_f32(int m, const float* M, const float* bias, const float* residual, int ldr,
                                                     float* out, int ldo, int flags, int N, int H, int W, int Cout, void* stats0,
                                                     int cpg0, int coff0, void* stats1, int cpg1, int coff1, int splits,
                                                     void* stream) {
    BBDM_WINO_M78(m);
    BBDM_REQUIRE(splits >= 1 && (splits == 1 || m < 6), "winograd_output: splits=%d (m = 6 / 7 / 8 layers are never split)", splits);
    BBDM_REQUIRE(m != 7 || ((flags & BBDM_CONV_OUT_PHASES) && Cout % 128 == 0),
                 "winograd_output: m = 7 is the phase-filter form (BBDM_CONV_OUT_PHASES, Cout %% 128 == 0)");
    BBDM_REQUIRE(M && out && N > 0, "winograd_output: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    BBDM_REQUIRE(Cout > 0 && Cout % 4 == 0 && ldo % 4 == 0 && ldo >= Cout, "winograd_output: Cout=%d ldo=%d", Cout, ldo);
    BBDM_REQUIRE((flags & ~(BBDM_CONV_RES_PER_IMAGE | BBDM_CONV_RES_UPSAMPLE | BBDM_CONV_OUT_PHASES)) == 0 &&
                     (flags & (BBDM_CONV_RES_PER_IMAGE | BBDM_CONV_RES_UPSAMPLE)) != (BBDM_CONV_RES_PER_IMAGE | BBDM_CONV_RES_UPSAMPLE),
                 "winograd_output: unsupported flags 0x%x", flags);
    const int ph = (flags & BBDM_CONV_OUT_PHASES) ? 1 : 0;
    BBDM_REQUIRE(!ph || (!residual && splits == 1), "winograd_output: BBDM_CONV_OUT_PHASES takes neither a residual nor split-K partials");
    const int Cm = ph ? 4 * Cout : Cout;                     // channels of M
    BBDM_REQUIRE(!(flags & BBDM_CONV_RES_UPSAMPLE) || (residual && H % 2 == 0 && W % 2 == 0),
                 "winograd_output: BBDM_CONV_RES_UPSAMPLE needs a residual and even H, W");
    BBDM_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= Cout && ((uintptr_t)residual & 15) == 0), "winograd_output: ldr=%d", ldr);
    BBDM_REQUIRE(!bias || ((uintptr_t)bias & 15) == 0, "winograd_output: bias alignment");
    BBDM_REQUIRE((((uintptr_t)M | (uintptr_t)out) & 15) == 0, "winograd_output: 16-byte alignment");
    BBDM_REQUIRE((!stats0 || (cpg0 > 0 && cpg0 % 4 == 0 && coff0 >= 0 && (coff0 + Cout - 1) / cpg0 < 32)) &&
                     (!stats1 || (cpg1 > 0 && cpg1 % 4 == 0 && coff1 >= 0 && (coff1 + Cout - 1) / cpg1 < 32)),
                 "winograd_output: statistics targets need cpg %% 4 == 0 and (coff + Cout) / cpg <= 32");
    StatArgs st;
    st.s[0] = (unsigned long long*)stats0; st.cpg[0] = cpg0 > 0 ? cpg0 : 1; st.coff[0] = coff0;
    st.s[1] = (unsigned long long*)stats1; st.cpg[1] = cpg1 > 0 ? cpg1 : 1; st.coff[1] = coff1;
    const size_t T = tiles_raw(N, H, W, m), Tp = tiles_padded(N, H, W, m);
    const long long units = (long long)T * (Cm / (m == 6 ? 2 : 4));
    // contiguous runs of iters x 256 units per workgroup: >= ~6000 workgroups (the chip holds ~1000 at a time: a 1536-workgroup
    // launch measured 13 % slower for its ragged last wave), <= 16 iterations
    long long iters = units / (256ll * 6144);
    iters = iters < 1 ? 1 : (iters > 16 ? 16 : iters);
    const long long blocks = (units + 256 * iters - 1) / (256 * iters);
    BBDM_REQUIRE(blocks < (1ll << 31), "winograd_output: too many workgroups");
    const int rpi = (flags & BBDM_CONV_RES_PER_IMAGE) ? 1 : (flags & BBDM_CONV_RES_UPSAMPLE) ? 2 : 0;     // residual addressing mode
    const dim3 g((unsigned)blocks), b(256);
    hipStream_t s_ = (hipStream_t)stream;
    // the two-phase LDS kernel wherever the channel blocks allow it; other shapes keep the one-thread-per-window kernels
    if (Cm % 128 == 0 && (!ph || Cout % 128 == 0) && (long long)T * (Cm / 128) < (1ll << 31)) {
        // Tiles per workgroup.  Two (double-buffered) by default; MORE where the layer is narrow: a 128-channel block of a layer with
        // 4 channels per group touches 32 groups = 64 accumulator cells, i.e. >= 128 integer atomics per workgroup whatever its tile
        // count -- at two tiles that was one atomic per 72 stored values and cost the 128-channel layers of the 256^2 level 20 %
        // (measured round 4).  8 cells per tile is the budget; a run of tiles must not span more than two images (the kernel keeps two
        // register accumulators) and the launch keeps >= 2048 workgroups.
        int tpw = 2;
        if (stats0 || stats1) {
            int cells = 0;
            if (stats0) cells += 2 * (128 / st.cpg[0] + 1);
            if (stats1) cells += 2 * (128 / st.cpg[1] + 1);
            tpw = cells / 8;
            if (tpw < 2) tpw = 2;
            if (tpw > 8) tpw = 8;
            while (tpw > 2 && ((long long)((T + tpw - 1) / tpw) * (Cm / 128) < 2048)) --tpw;
        }
        const long long per_image = (long long)wino_tdim(H, m) * wino_tdim(W, m);
        if (tpw > per_image) tpw = (int)per_image;
        if (tpw < 1) tpw = 1;
        if (m == 7) {
            if (tpw == 1)
                hipLaunchKernelGGL((winograd_output_lds_kernel<7, false, 1, 8>), dim3((unsigned)(T * (size_t)(Cm / 128))), dim3(512), 0, s_, M,
                                   Tp * (size_t)Cm, Cm, 1, bias, residual, ldr, rpi, out, ldo, N, H, W, Cout, Cm / 128, (long long)T, st, ph, 1);
            else
                hipLaunchKernelGGL((winograd_output_lds_kernel<7, false, 2, 8>), dim3((unsigned)(((T + tpw - 1) / tpw) * (size_t)(Cm / 128))),
                                   dim3(512), 0, s_, M, Tp * (size_t)Cm, Cm, 1, bias, residual, ldr, rpi, out, ldo, N, H, W, Cout, Cm / 128,
                                   (long long)T, st, ph, tpw);
            BBDM_CHECK_LAUNCH("winograd_output(m = 7)");
            return BBDM_OK;
        }
#define BBDM_WINO_OUT(MO_, RES_, TPW_)                                                                                              \
    hipLaunchKernelGGL((winograd_output_lds_kernel<MO_, RES_, TPW_>), dim3((unsigned)(((T + tpw - 1) / tpw) * (size_t)(Cm / 128))), \
                       dim3((MO_ + 2) * 64), 0, s_, M, Tp * (size_t)Cm, Cm, splits, bias, residual, ldr, rpi, out, ldo, N, H, W, Cout, \
                       Cm / 128, (long long)T, st, ph, tpw)
#define BBDM_WINO_OUT_M(MO_)                                                                                   \
    do {                                                                                                       \
        if (tpw == 1) { if (residual) BBDM_WINO_OUT(MO_, true, 1); else BBDM_WINO_OUT(MO_, false, 1); }        \
        else { if (residual) BBDM_WINO_OUT(MO_, true, 2); else BBDM_WINO_OUT(MO_, false, 2); }                 \
    } while (0)
        if (m == 8) {      // two buffers = 80 KB of (dynamic) LDS, two workgroups of ten waves per CU
            constexpr size_t lds8 = (size_t)2 * 8 * 10 * 64 * sizeof(float2);
            static bool attr_set_dev[BBDM_MAX_DEVICES] = {};
            bool& attr_set = attr_set_dev[bbdm_device_slot()];
            if (!attr_set) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_output_lds_kernel<8, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_output_lds_kernel<8, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) != hipSuccess) {
                    bbdm_set_error("winograd_output: hipFuncSetAttribute(%zu B LDS) failed", lds8);
                    return BBDM_E_LAUNCH;
                }
                attr_set = true;
            }
            if (tpw < 2) tpw = per_image >= 2 ? 2 : 1;
#define BBDM_WINO_OUT8(RES_)                                                                                                        \
    hipLaunchKernelGGL((winograd_output_lds_kernel<8, RES_, 2>), dim3((unsigned)(((T + tpw - 1) / tpw) * (size_t)(Cm / 128))), dim3(640), \
                       lds8, s_, M, Tp * (size_t)Cm, Cm, splits, bias, residual, ldr, rpi, out, ldo, N, H, W, Cout, Cm / 128, (long long)T, st, ph, tpw)
            if (residual) BBDM_WINO_OUT8(true); else BBDM_WINO_OUT8(false);
#undef BBDM_WINO_OUT8
        }
        else if (m == 6) BBDM_WINO_OUT_M(6); else if (m == 4) BBDM_WINO_OUT_M(4); else BBDM_WINO_OUT_M(2);
#undef BBDM_WINO_OUT_M
#undef BBDM_WINO_OUT
        BBDM_CHECK_LAUNCH("winograd_output");
        return BBDM_OK;
    }
    BBDM_REQUIRE(m != 7 && m != 8, "winograd_output: m = 7 / 8 need Cout %% 128 == 0 (the two-phase kernel)");
    if (m == 6 && residual)
        hipLaunchKernelGGL(winograd_output6_kernel<true>, g, b, 0, s_, M, Tp * (size_t)Cm, Cm, bias, residual, ldr, rpi, out,
                           ldo, N, H, W, Cout, (int)iters, st, ph);
    else if (m == 6)
        hipLaunchKernelGGL(winograd_output6_kernel<false>, g, b, 0, s_, M, Tp * (size_t)Cm, Cm, bias, residual, ldr, rpi, out,
                           ldo, N, H, W, Cout, (int)iters, st, ph);
    else if (m == 2)
        hipLaunchKernelGGL(winograd_output_kernel<2>, g, b, 0, s_, M, Tp * (size_t)Cm, Cm, splits, bias, residual, ldr, rpi, out,
                           ldo, N, H, W, Cout, (int)iters, st, ph);
    else
        hipLaunchKernelGGL(winograd_output_kernel<4>, g, b, 0, s_, M, Tp * (size_t)Cm, Cm, splits, bias, residual, ldr, rpi, out,
                           ldo, N, H, W, Cout, (int)iters, st, ph);
    BBDM_CHECK_LAUNCH("winograd_output");
    return BBDM_OK;
}

extern "C" int bbdm_winograd_output_stats_f32(int m, const float* M, const float* bias, const float* residual, int ldr,
                                              float* out, int ldo, int flags, int N, int H, int W, int Cout, void* stats0,
                                              int cpg0, int coff0, void* stats1, int cpg1, int coff1,
                                              void* stream) {
    return bbdm_winograd_output_splitk_stats_f32(m, M, bias, residual, ldr, out, ldo, flags, N, H, W, Cout, stats0, cpg0, coff0,
                                                 stats1, cpg1, coff1, 1, stream);
}

extern "C" int bbdm_winograd_output_f32(int m, const float* M, const float* bias, const float* residual, int ldr,
                                        float* out, int ldo, int flags, int N, int H, int W, int Cout, void* stream) {
    return bbdm_winograd_output_stats_f32(m, M, bias, residual, ldr, out, ldo, flags, N, H, W, Cout, nullptr, 0, 0, nullptr, 0,
                                          0, stream);
}

extern "C" int bbdm_conv3x3_winograd_f32(int m, const float* x, int ldx, const float* packed_wino, const float* bias,
                                         const float* residual, int ldr, float* out, int ldo, int flags, float* ws, int N,
                                         int H, int W, int CinPad, int Cout, void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(ws, "winograd: null workspace");
    BBDM_REQUIRE(N > 0 && H > 0 && W > 0 && CinPad > 0, "winograd: bad shape");
    float* V = ws;                                                                    // [(m+2)^2][tiles][CinPad]
    float* M = ws + (size_t)planes(m) * tiles_padded(N, H, W, m) * CinPad;            // [(m+2)^2][tiles][Cout]
    int rc = bbdm_winograd_input_f32(m, x, ldx, V, nullptr, nullptr, 0, 0, 0, N, H, W, CinPad, stream);
    if (rc == BBDM_OK) rc = bbdm_winograd_gemm_f32(m, V, packed_wino, M, N, H, W, CinPad, Cout, stream);
    if (rc == BBDM_OK) rc = bbdm_winograd_output_f32(m, M, bias, residual, ldr, out, ldo, flags, N, H, W, Cout, stream);
    return rc;
}

// dY [N,H,W,Cout] (pitch ld) -> dMt = the transposed bf16 planes of A dY A^T (bbdm_gemm_bf3p_tn_bt_bytes((m+2)^2, tiles, Cout) bytes)
// and dm11 [tiles][Cout] fp32 = its plane (1, 1), whose column sums are the bias gradient.  Cout a multiple of 4; the channels up to
// the next multiple of 32 (whole fragment units) are written as zeros.
static int winograd_dy_planes(int m, const float* dy, int ld, void* dMt, float* dm11, int N, int H, int W, int Cout, const float* dybound,
                              void* stream);
extern "C" int bbdm_winograd_dy_transform_bf3p_f32(int m, const float* dy, int ld, void* dMt, float* dm11, int N, int H, int W,
                                                   int Cout, void* stream) {
    return winograd_dy_planes(m, dy, ld, dMt, dm11, N, H, W, Cout, nullptr, stream);
}
// ... on the fp16 pair (dMt: bbdm_gemm_h2p_tn_bt_bytes bytes) under dybound = a device float >= max |dY| (bbdm_absmax_rows_f32)
extern "C" int bbdm_winograd_dy_transform_h2p_f32(int m, const float* dy, int ld, void* dMt, float* dm11, int N, int H, int W,
                                                  int Cout, const float* dybound, void* stream) {
    BBDM_REQUIRE(dybound, "winograd_dy_h2p: null bound");
    return winograd_dy_planes(m, dy, ld, dMt, dm11, N, H, W, Cout, dybound, stream);
}
extern "C" float bbdm_winograd_dy_gain(int m) { return wino_dy_gain(m); }
static int winograd_dy_planes(int m, const float* dy, int ld, void* dMt, float* dm11, int N, int H, int W, int Cout, const float* dybound,
                              void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(dy && dMt && dm11 && N > 0, "winograd_dy_bf3p: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    BBDM_REQUIRE(Cout > 0 && Cout % 2 == 0 && ld % 2 == 0 && ld >= Cout && ((uintptr_t)dy & 7) == 0 && ((uintptr_t)dMt & 15) == 0 &&
                     ((uintptr_t)dm11 & 7) == 0, "winograd_dy_bf3p: Cout=%d ld=%d / alignment", Cout, ld);
    const size_t T = tiles_raw(N, H, W, m), Tp = tiles_padded(N, H, W, m);
    const int CoutPad = cdiv(Cout, 128) * 128;                   // the B operand of the GEMM: whole 128-column tiles
    const int nchunks = CoutPad / KC, TG = (int)(Tp / 8);
    const size_t plane_t = (size_t)CoutPad * Tp * (dybound ? 4 : 6);
    const long long blocks = 8ll * ((TG + 7) / 8) * nchunks;
    BBDM_REQUIRE(blocks < (1ll << 31), "winograd_dy_bf3p: too many workgroups");
    const dim3 g((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
#define BBDM_WINO_DYS(MO, NPL)                                                                                               \
    hipLaunchKernelGGL((winograd_dy_split_kernel<MO, NPL>), g, dim3((MO + 2) * 64), 0, st, dy, ld, (unsigned char*)dMt, dm11, N, H, W, \
                       Cout, nchunks, (long long)T, TG, plane_t, (int)(Tp / 16), dybound)
    if (dybound) { if (m == 2) BBDM_WINO_DYS(2, 2); else if (m == 4) BBDM_WINO_DYS(4, 2); else if (m == 8) BBDM_WINO_DYS(8, 2); else BBDM_WINO_DYS(6, 2); }
    else if (m == 2) BBDM_WINO_DYS(2, 3); else if (m == 4) BBDM_WINO_DYS(4, 3); else if (m == 8) BBDM_WINO_DYS(8, 3); else BBDM_WINO_DYS(6, 3);
#undef BBDM_WINO_DYS
    BBDM_CHECK_LAUNCH("winograd_dy_bf3p");
    return BBDM_OK;
}

// Test hook (exported, not part of the public header): the 1-D transforms above evaluated on the HOST, so that the CPU
// test-suite can check the hand-factored formulas against the transform matrices (tests/test_winograd_math_cpu.py).
// which: 0 = B^T (m+2 -> m+2), 1 = A^T (m+2 -> m), 2 = G (3 -> m+2), 3 = A (m -> m+2), 4 = G^T (m+2 -> 3).
extern "C" int bbdm_debug_winograd_transform_1d(int m, int which, const float* in, float* out) {
    BBDM_WINO_M78(m);
    BBDM_REQUIRE(in && out && which >= 0 && which <= 4, "winograd_transform_1d: bad args");
    if (m == 8) {                      // F(8x8, 3x3): G / G^T in fp64 (as the weight / finish kernels evaluate them)
        if (which == 3) {
            float v[8], r[10];
            for (int i = 0; i < 8; ++i) v[i] = in[i];
            a_transform<8>(v, r);
            for (int i = 0; i < 10; ++i) out[i] = r[i];
        } else if (which == 4) {
            double u[10], g[3];
            for (int i = 0; i < 10; ++i) u[i] = in[i];
            gt_transform<8>(u, g);
            for (int i = 0; i < 3; ++i) out[i] = (float)g[i];
        } else if (which == 0) {
            float d[10], t[10];
            for (int i = 0; i < 10; ++i) d[i] = in[i];
            bt_transform<8>(d, t);
            for (int i = 0; i < 10; ++i) out[i] = t[i];
        } else if (which == 1) {
            float v[10], r[8];
            for (int i = 0; i < 10; ++i) v[i] = in[i];
            at_transform<8>(v, r);
            for (int i = 0; i < 8; ++i) out[i] = r[i];
        } else {
            double g[3] = {in[0], in[1], in[2]}, u[10];
            g_transform<8>(g, u);
            for (int i = 0; i < 10; ++i) out[i] = (float)u[i];
        }
        return BBDM_OK;
    }
    if (m == 7) {                      // F(7x7, 2x2): B^T is m = 6's (8 -> 8), A^T 8 -> 7, G 2 -> 8
        BBDM_REQUIRE(which <= 2, "winograd_transform_1d: m = 7 has no weight-gradient side");
        if (which == 0) {
            float d[8], t[8];
            for (int i = 0; i < 8; ++i) d[i] = in[i];
            bt_transform<6>(d, t);
            for (int i = 0; i < 8; ++i) out[i] = t[i];
        } else if (which == 1) {
            float v[8], r[7];
            for (int i = 0; i < 8; ++i) v[i] = in[i];
            at_transform7(v, r);
            for (int i = 0; i < 7; ++i) out[i] = r[i];
        } else {
            float g[2] = {in[0], in[1]}, u[8];
            g_transform72(g, u);
            for (int i = 0; i < 8; ++i) out[i] = u[i];
        }
        return BBDM_OK;
    }
#define BBDM_WINO_1D(MO)                                                             \
    do {                                                                             \
        if (which == 0) {                                                            \
            float d[MO + 2], t[MO + 2];                                              \
            for (int i = 0; i < MO + 2; ++i) d[i] = in[i];                           \
            bt_transform<MO>(d, t);                                                  \
            for (int i = 0; i < MO + 2; ++i) out[i] = t[i];                          \
        } else if (which == 1) {                                                     \
            float v[MO + 2], r[MO];                                                  \
            for (int i = 0; i < MO + 2; ++i) v[i] = in[i];                           \
            at_transform<MO>(v, r);                                                  \
            for (int i = 0; i < MO; ++i) out[i] = r[i];                              \
        } else if (which == 2) {                                                     \
            float g[3], u[MO + 2];                                                   \
            for (int i = 0; i < 3; ++i) g[i] = in[i];                                \
            g_transform<MO>(g, u);                                                   \
            for (int i = 0; i < MO + 2; ++i) out[i] = u[i];                          \
        } else if (which == 3) {                                                     \
            float v[MO], r[MO + 2];                                                  \
            for (int i = 0; i < MO; ++i) v[i] = in[i];                               \
            a_transform<MO>(v, r);                                                   \
            for (int i = 0; i < MO + 2; ++i) out[i] = r[i];                          \
        } else {                                                                     \
            float u[MO + 2], g[3];                                                   \
            for (int i = 0; i < MO + 2; ++i) u[i] = in[i];                           \
            gt_transform<MO>(u, g);                                                  \
            for (int i = 0; i < 3; ++i) out[i] = g[i];                               \
        }                                                                            \
    } while (0)
    if (m == 2) BBDM_WINO_1D(2); else if (m == 4) BBDM_WINO_1D(4); else BBDM_WINO_1D(6);
#undef BBDM_WINO_1D
    return BBDM_OK;
}
