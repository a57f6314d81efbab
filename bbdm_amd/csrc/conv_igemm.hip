// conv_igemm.hip -- fp32 NHWC convolution (3x3 pad 1 / 1x1, stride 1) as an implicit GEMM on the CDNA4
// f32-input matrix core (v_mfma_f32_32x32x2_f32, exact fp32 = an fmaf chain).
//
// Replaces the reference's nn.Conv2d / nn.Conv1d(k=1) call sites: openaimodel.py:207,233,244,307,315,524,690.
//
// GEMM view:  out[m, co] = sum_{tap, ci} x[pixel(m) + tap, ci] * W[tap, ci, co],   m = (n, h, w)
//
// Work decomposition (one workgroup = 4 or 8 waves = BM output pixels x BN output channels):
//   * the BM pixels are IMGS whole-or-partial images x a TH x TW spatial tile (all powers of two), so that the
//     input needed by all 9 taps is ONE halo patch (TH+2) x (TW+2) per image, staged into LDS once per
//     16-channel chunk and re-used by the 9 taps (9x fewer L2/HBM reads than a materialised im2col).
//   * K is walked chunk-major, tap-minor ("phase" = (chunk, tap)); per phase the [BN x 16] weight slab is
//     staged (the packed layout makes it one contiguous 8 KB run), double-buffered; the patch is
//     double-buffered per chunk.  Global loads for phase p+1 are issued before phase p's MFMAs and written to
//     the other LDS buffer after them: one __syncthreads per phase, HBM/L2 latency hidden under 4096 cycles
//     of MFMA per wave.
//   * LDS rows are [pixel][16 ch + 4 pad] and [cout][16 ch + 4 pad]: a lane fetches 4 consecutive k with one
//     ds_read_b128 (lanes 0-31: k0..k0+3, lanes 32-63: k0+4..k0+7, matching the 32x32x2 operand map where
//     lanes>=32 carry the second k) and the 20-float pitch makes every 16-lane b128 group hit 16 distinct
//     16-byte slots -> conflict-free.
//   * every wave owns a 64x64 output tile (4 accumulators of 16 VGPRs); the 256x128 block runs 8 waves and two
//     blocks share a CU (<=75 KB LDS, <=128 VGPR each): 4 waves per SIMD cover each other's barriers.
#include <stdlib.h>
#include "common.h"
#include "stats_acc.h"

namespace {

constexpr int KC = 16;   // input channels per chunk (also the packed-weight inner dimension)
constexpr int KP = 20;   // LDS pitch of one pixel / one cout row (16 + 4 pad floats)

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* out;
    int ldx, ldr, ldo, out_nchw;
    int N, H, W, Cin, Cout, CoutPad;
    int taps, nchunks, pad;
    int TWl, THl, imgs;         // log2 of tile width / height; images per block (<= BM / (TH*TW))
    int PW, PH, patchPix;       // patch geometry (incl. halo), pixels per block patch
    int tilesX, tilesY, tilesN;
    int splits, chunks_per_split;   // split-K over the Cin chunks (gridDim.y); partial sums go to ws
    float* ws;                      // [splits][N*H*W][ldw]
    size_t ws_cap;                  // capacity of ws in floats
    int ldw;
    // optional fused producer: x is replaced by act(x * pre_sc[n][c] + pre_bi[n][c]) while it is staged (the
    // GroupNorm affine [+ FiLM] [+ SiLU] of the reference, with per-image coefficients from bbdm_groupnorm_coeffs_f32)
    const float* pre_sc;
    const float* pre_bi;
    int pre_ld, pre_silu;
    // batched GEMM mode (gridDim.z = batch): per-batch element offsets of x / packed weights / out.  Used by the
    // Winograd path (16 transformed-domain GEMMs in one launch); bias, residual and split-K are off in this mode.
    int batch;
    size_t xz, wz, oz;
    // GroupNorm statistics of the output accumulated by this launch (VAR bit 3; see winograd.hip: StatArgs): the exact limb
    // accumulators [N][32][2][SA_W] 64-bit words (stats_acc.h; bbdm_groupnorm_stats_bytes(N, 32) bytes each) of up to two consumers,
    // their group width and the channel offset of `out` in their tensor
    unsigned long long* st_s[2];
    int st_cpg[2], st_coff[2];
};

// ---- epilogue: + bias (+ residual) -> global ----------------------------------------------------------------------
// MODE 0: NHWC, no residual   1: NHWC + per-pixel residual   2: NHWC + per-image residual row   3: NCHW (the head)
// 4: split-K partial sums.  The mode is a compile-time parameter because gfx9 counts loads AND stores in the one
// in-order vmcnt: with a run-time `if (residual)` inside the row loop the compiler waits vmcnt(0) in front of every
// element, i.e. for the previous store to complete -- 64 serialised stores = 38 us per tile (measured with per-workgroup
// timestamps), 12 % of a K = 1024 tile.  Mode 0 issues nothing but stores; the residual modes fetch the residuals of
// 8 rows, wait once, then store.
// Row loop outside, column tiles inside: the pixel index of a row is computed once and consumed at once (with the column
// tile outermost all MT*16 row addresses stayed live and accumulators were spilled to scratch).
template <int BM, int BN, int WM, int WN, int MODE, bool STATS = false>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& a, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int tile_x,
                                                   int tile_y, int img0, int cout0, int wm, int wn, int lane,
                                                   double* psum = nullptr, double* psq = nullptr) {
    constexpr int MT = BM / WM / 32, NTL = BN / WN / 32;
    constexpr int RG = 8;                        // rows per residual batch
    const int TW = 1 << a.TWl, TH = 1 << a.THl;
    const int hbase = tile_y * TH, wbase = tile_x * TW;
    int co[NTL];
    bool cok[NTL];
    float bv[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        co[nt] = cout0 + wn * (BN / WN) + nt * 32 + (lane & 31);
        cok[nt] = co[nt] < a.Cout;
        bv[nt] = (MODE != 4 && cok[nt] && a.bias) ? a.bias[co[nt]] : 0.0f;
    }
    // One explicit wait on the path that dominates every store: without it the waitcnt pass, unable to prove across the
    // per-row branches that the bias load has landed, re-inserts vmcnt(0) in front of each store.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) as a real S_WAITCNT the waitcnt pass accounts for
    auto row_of = [&](int mt, int r, int& n, int& h, int& w) -> bool {
        const int m = wm * (BM / WM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int img_l = m >> (a.THl + a.TWl);
        n = img0 + img_l;
        h = hbase + ((m >> a.TWl) & (TH - 1));
        w = wbase + (m & (TW - 1));
        return img_l < a.imgs && n < a.N && h < a.H && w < a.W;
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RG) {
            float rv[RG][NTL];
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int rr = 0; rr < RG; ++rr) {
                    int n, h, w;
                    const bool ok = row_of(mt, r0 + rr, n, h, w);
                    const size_t roff = MODE == 2 ? (size_t)n * a.ldr : ((size_t)(n * a.H + h) * a.W + w) * a.ldr;
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt) rv[rr][nt] = (ok && cok[nt]) ? a.res[roff + co[nt]] : 0.0f;
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);      // the batch has landed (same reason as above)
            }
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
                const int r = r0 + rr;
                int n, h, w;
                if (!row_of(mt, r, n, h, w)) continue;
                const size_t pix = (size_t)(n * a.H + h) * a.W + w;
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    if (!cok[nt]) continue;
                    float v = acc[mt][nt][r] + bv[nt];
                    if (MODE == 1 || MODE == 2) v += rv[rr][nt];
                    if (MODE == 4)          // raw partial sum; bias / residual are applied by the reduce kernel
                        a.ws[((size_t)blockIdx.y * a.N * a.H * a.W + pix) * a.ldw + co[nt]] = acc[mt][nt][r];
                    else if (MODE == 3) {
                        if (a.res) v += (a.out_nchw & 2) ? a.res[(size_t)n * a.ldr + co[nt]] : a.res[pix * a.ldr + co[nt]];
                        a.out[((size_t)(n * a.Cout + co[nt]) * a.H + h) * a.W + w] = v;
                    } else {
                        a.out[pix * a.ldo + co[nt]] = v;
                        if (STATS) { psum[nt] += (double)v; psq[nt] += (double)v * v; }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the row groups sequential: no batching of address math
        }
    }
}

// Epilogue + GroupNorm statistics of the stored values (modes 0-2, one image per workgroup: the host checks that): per-lane
// fp64 partials -> exact limb accumulators in LDS (the pipeline's buffers are idle by now) -> one cell add per touched (image, group).
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_stats(const ConvArgs& a, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int tile_x,
                                                    int tile_y, int img0, int cout0, int wm, int wn, int lane, float* smem) {
    constexpr int NTL = BN / WN / 32;
    double psum[NTL], psq[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) psum[nt] = psq[nt] = 0.0;
    if (!a.res)
        conv_epilogue_rows<BM, BN, WM, WN, 0, true>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane, psum, psq);
    else if (a.out_nchw & 2)
        conv_epilogue_rows<BM, BN, WM, WN, 2, true>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane, psum, psq);
    else
        conv_epilogue_rows<BM, BN, WM, WN, 1, true>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane, psum, psq);
    // exact limb accumulators (stats_acc.h: integer atomics, order-independent) in LDS, then one cell per touched (image, group)
    unsigned long long* ls = reinterpret_cast<unsigned long long*>(smem);            // [2 consumers][32 groups][2][SA_W]
    const int tid = threadIdx.x;
    for (int i = tid; i < 128 * SA_W; i += WM * WN * 64) ls[i] = 0ull;
    __syncthreads();
    // lanes l and l + 32 hold the same channel (other rows): add the halves, then the lanes of a group (consecutive channels) meet in
    // the group's last lane by a segmented scan -- one LDS cell add per group instead of one per lane (same-address LDS atomics serialise)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int co = cout0 + wn * (BN / WN) + nt * 32 + (lane & 31);
        const bool live = co < a.Cout;
        const double hs = (live ? psum[nt] : 0.0) + __shfl_xor(live ? psum[nt] : 0.0, 32);
        const double hq = (live ? psq[nt] : 0.0) + __shfl_xor(live ? psq[nt] : 0.0, 32);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!a.st_s[k]) continue;
            const int g = live ? (a.st_coff[k] + co) / a.st_cpg[k] : -1;
            double vs = hs, vq = hq;
            sa_seg_scan2(vs, vq, g, lane, a.st_cpg[k] < 32 ? a.st_cpg[k] : 32, 32);
            const bool tail = sa_seg_tail(g, lane, 32);            // (a cross-lane op: every lane of the wave executes it)
            if (lane < 32 && live && tail) {
                sa_add(ls + (size_t)((k * 32 + g) * 2) * SA_W, vs);
                sa_add(ls + (size_t)((k * 32 + g) * 2 + 1) * SA_W, vq);
            }
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int k = tid >> 6;
        if (a.st_s[k]) sa_add_cell(a.st_s[k] + ((size_t)img0 * 64 + (tid & 63)) * SA_W, ls + (size_t)tid * SA_W);
    }
}

template <int BM, int BN, int WM, int WN, bool SPLIT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int tile_x,
                                              int tile_y, int img0, int cout0, int wm, int wn, int lane) {
    if (SPLIT)
        conv_epilogue_rows<BM, BN, WM, WN, 4>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane);
    else if (a.out_nchw & 1)
        conv_epilogue_rows<BM, BN, WM, WN, 3>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane);
    else if (!a.res)
        conv_epilogue_rows<BM, BN, WM, WN, 0>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane);
    else if (a.out_nchw & 2)
        conv_epilogue_rows<BM, BN, WM, WN, 2>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane);
    else
        conv_epilogue_rows<BM, BN, WM, WN, 1>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane);
}

// ---- OIHW -> packed [tap][chunk][CoutPad][16] -------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int CinPad,
                                   int ks, int CoutPad, int nchunks) {
    const size_t total = (size_t)ks * ks * nchunks * CoutPad * KC;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = i % KC;
        size_t t = i / KC;
        const int co = t % CoutPad;
        t /= CoutPad;
        const int chunk = t % nchunks;
        const int tap = t / nchunks;
        const int ci = chunk * KC + k;
        float v = 0.0f;
        if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * ks * ks + tap];
        p[i] = v;
    }
}

// VAR bit 0: fused producer (pre_sc / pre_bi) present; bit 1: split-K epilogue (raw partial sums to the workspace).
// The common case (VAR = 0) carries neither branch, so its register allocation is that of the plain kernel.
// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2) in linear launch order.  Give every XCD a
// CONTIGUOUS range of the (n_tile fastest) tile order instead, so that the Cout tiles of one pixel tile -- which read the
// same input rows -- run on the same XCD and share them through its L2 rather than each fetching them over the fabric.
__device__ __forceinline__ int xcd_swizzled_block() {
    const int nblk = (int)gridDim.x, x = (int)blockIdx.x;
    if (nblk < 64) return x;
    const int off = (int)(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * (size_t)nblk % 8);   // linear id of x = 0, mod 8
    const int c = (off + x) & 7;                 // the XCD this workgroup lands on
    int start = 0;
    for (int cc = 0; cc < 8; ++cc) {
        if (cc == c) break;
        const int first = (cc - off) & 7;        // smallest x on XCD cc
        start += (nblk - first + 7) >> 3;
    }
    return start + ((x - ((c - off) & 7)) >> 3);
}

template <int BM, int BN, int WM, int WN, int PSLOTS, int OCC, int VAR>
__global__ void __launch_bounds__(WM * WN * 64, OCC * WM * WN / 4)
conv_igemm_f32(const ConvArgs a_in) {
    constexpr bool PRE = (VAR & 1) != 0, SPLIT = (VAR & 2) != 0, GEMM = (VAR & 4) != 0, STATS = (VAR & 8) != 0;
    ConvArgs a = a_in;
    a.x += (size_t)blockIdx.z * a.xz;
    a.w += (size_t)blockIdx.z * a.wz;
    a.out += (size_t)blockIdx.z * a.oz;
    constexpr int NTHR = WM * WN * 64;
    constexpr int MT = BM / WM / 32;     // 32-row MFMA tiles per wave (M)
    constexpr int NTL = BN / WN / 32;    // 32-col MFMA tiles per wave (N)
    constexpr int WSLOTS = (BN * KC / 4) / NTHR;
    static_assert((BN * KC / 4) % NTHR == 0, "weight slab must divide evenly");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int patchFloats = a.patchPix * KP;
    float* pbuf = smem;                          // [2][patchPix][KP]
    float* wbuf = smem + 2 * patchFloats;        // [2][BN][KP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- block -> tile ------------------------------------------------------------------------------
    int bid = xcd_swizzled_block();
    const int n_tile = bid % a.tilesN;
    bid /= a.tilesN;
    const int tile_x = bid % a.tilesX;
    bid /= a.tilesX;
    const int tile_y = bid % a.tilesY;
    const int ig = bid / a.tilesY;
    const int TW = 1 << a.TWl, TH = 1 << a.THl;
    const int img0 = ig * a.imgs;
    const int h0 = tile_y * TH - a.pad, w0 = tile_x * TW - a.pad;
    const int cout0 = n_tile * BN;

    // ---- per-thread patch staging slots (same pixels for every chunk) ---------------------------------
    uint32_t goff[PSLOTS];
    uint32_t pvalid = 0;
    uint32_t pimg = 0;                 // 5 bits per slot: image (relative to img0) the slot's pixel belongs to
    const int nPatchVec = a.patchPix * (KC / 4);
#pragma unroll
    for (int s = 0; s < PSLOTS; ++s) {
        const int f = tid + s * NTHR;
        goff[s] = 0;
        if (f < nPatchVec) {
            const int pp = f >> 2, c4 = f & 3;
            const int img_l = pp / (a.PH * a.PW);
            const int rem = pp - img_l * (a.PH * a.PW);
            const int py = rem / a.PW, px = rem - py * a.PW;
            const int n = img0 + img_l, h = h0 + py, w = w0 + px;
            if (n < a.N && h >= 0 && h < a.H && w >= 0 && w < a.W) {
                goff[s] = (uint32_t)(((n * a.H + h) * a.W + w)) * (uint32_t)a.ldx + c4 * 4;
                pvalid |= 1u << s;
                if (PRE) pimg |= (uint32_t)(img_l & 31) << (5 * s);
            }
        }
    }

    // ---- per-lane fragment bases --------------------------------------------------------------------
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = wm * (BM / WM) + mt * 32 + (lane & 31);
        int img_l = m >> (a.THl + a.TWl);
        if (img_l >= a.imgs) img_l = 0;      // padding rows of the M tile: read something valid, discarded later
        const int ph = (m >> a.TWl) & (TH - 1);
        const int pw = m & (TW - 1);
        abase[mt] = (img_l * a.PH * a.PW + ph * a.PW + pw) * KP + (lane >> 5) * 4;
    }
    int bbase[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) bbase[nt] = (wn * (BN / WN) + nt * 32 + (lane & 31)) * KP + (lane >> 5) * 4;

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const size_t wPhaseStride = (size_t)a.CoutPad * KC;
    const float* wsrc = a.w + (size_t)cout0 * KC + tid * 4;   // + phase * wPhaseStride + s * NTHR * 4

    float4 preg[PSLOTS];
    float4 wreg[WSLOTS];

    auto load_patch_to = [&](float4 (&preg)[PSLOTS], int chunk) {
        const int cbase = chunk * KC;
#pragma unroll
        for (int s = 0; s < PSLOTS; ++s) {
            const int c = cbase + ((tid + s * NTHR) & 3) * 4;
            if (((pvalid >> s) & 1u) && c < a.Cin)
                preg[s] = *reinterpret_cast<const float4*>(a.x + goff[s] + cbase);
            else
                preg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch_from = [&](float4 (&preg)[PSLOTS], float* dst) {
#pragma unroll
        for (int s = 0; s < PSLOTS; ++s) {
            const int f = tid + s * NTHR;
            if (f < nPatchVec) *reinterpret_cast<float4*>(dst + (f >> 2) * KP + (f & 3) * 4) = preg[s];
        }
    };
    auto load_patch = [&](int chunk) { load_patch_to(preg, chunk); };
    auto store_patch = [&](float* dst, int chunk) {
        if (PRE) {
            // fused GroupNorm affine (+ FiLM) (+ SiLU) on the valid elements; padding stays exactly zero
            const int cbase = chunk * KC;
#pragma unroll
            for (int s = 0; s < PSLOTS; ++s) {
                const int c = cbase + ((tid + s * NTHR) & 3) * 4;
                if (((pvalid >> s) & 1u) && c < a.Cin) {
                    const size_t o = (size_t)(img0 + ((pimg >> (5 * s)) & 31u)) * a.pre_ld + c;
                    const float4 sc = *reinterpret_cast<const float4*>(a.pre_sc + o);
                    const float4 bi = *reinterpret_cast<const float4*>(a.pre_bi + o);
                    float4 v = preg[s];
                    v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
                    if (a.pre_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                    preg[s] = v;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < PSLOTS; ++s) {
            const int f = tid + s * NTHR;
            if (f < nPatchVec) *reinterpret_cast<float4*>(dst + (f >> 2) * KP + (f & 3) * 4) = preg[s];
        }
    };
    auto load_w = [&](int phase) {
        // packed order is [tap][chunk][CoutPad][KC]; phase = chunk * taps + tap
        const int chunk = phase / a.taps, tap = phase - chunk * a.taps;
        const float* src = wsrc + (size_t)(tap * a.nchunks + chunk) * wPhaseStride;
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) wreg[s] = *reinterpret_cast<const float4*>(src + s * NTHR * 4);
    };
    auto store_w = [&](float* dst) {
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int f = tid + s * NTHR;
            *reinterpret_cast<float4*>(dst + (f >> 2) * KP + (f & 3) * 4) = wreg[s];
        }
    };

    // ---- prologue ------------------------------------------------------------------------------------
    const int chunk_begin = blockIdx.y * a.chunks_per_split;
    const int chunk_end = min(a.nchunks, chunk_begin + a.chunks_per_split);
    const int phase_begin = chunk_begin * a.taps;
    load_patch(chunk_begin);
    load_w(phase_begin);
    store_patch(pbuf + (chunk_begin & 1) * patchFloats, chunk_begin);
    store_w(wbuf + (phase_begin & 1) * (BN * KP));
    __syncthreads();

    const int nphase = chunk_end * a.taps;
    int chunk = chunk_begin, tap = 0;
    auto mfma_phase = [&](const float* P, const float* Wb) {
#pragma unroll
        for (int kg = 0; kg < KC / 8; ++kg) {
            float4 af[MT], bf[NTL];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const float4*>(P + abase[mt] + kg * 8);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) bf[nt] = *reinterpret_cast<const float4*>(Wb + bbase[nt] + kg * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].x, bf[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].y, bf[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].z, bf[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].w, bf[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
    };
    if constexpr (GEMM) {
        // One tap per chunk (plain GEMM, the Winograd tile GEMMs): every phase needs a fresh activation tile straight
        // from HBM, and one phase of compute (~3.4 us with two workgroups per CU) does not cover that latency under
        // load -- measured 4.6 us per phase.  The activation tile is therefore requested TWO phases ahead (two register
        // sets, loop unrolled by two so that they keep static names); the weight slab (L2-resident) stays one ahead.
        float4 preg2[PSLOTS];
        auto gemm_phase = [&](int phase, float4 (&cur)[PSLOTS], float4 (&nxt)[PSLOTS]) {
            // cur holds chunk phase+1 (requested during the previous phase); nxt receives chunk phase+2
            if (phase + 1 < nphase) load_w(phase + 1);
            if (phase + 2 < nphase) load_patch_to(nxt, phase + 2);
            mfma_phase(pbuf + (phase & 1) * patchFloats, wbuf + (phase & 1) * (BN * KP));
            if (phase + 1 < nphase) {
                store_w(wbuf + ((phase + 1) & 1) * (BN * KP));
                store_patch_from(cur, pbuf + ((phase + 1) & 1) * patchFloats);
            }
            __syncthreads();
        };
        if (phase_begin + 1 < nphase) load_patch_to(preg, phase_begin + 1);
        for (int phase = phase_begin; phase < nphase; phase += 2) {
            gemm_phase(phase, preg, preg2);
            if (phase + 1 < nphase) gemm_phase(phase + 1, preg2, preg);
        }
    } else
    for (int phase = phase_begin; phase < nphase; ++phase) {
        const bool has_next = phase + 1 < nphase;
        const bool last_tap = tap == a.taps - 1;
        if (has_next) load_w(phase + 1);
        if (last_tap && has_next) load_patch(chunk + 1);

        const int r = tap / 3, s = tap - r * 3;
        const float* P = pbuf + (chunk & 1) * patchFloats + (a.taps == 1 ? 0 : (r * a.PW + s) * KP);
        const float* Wb = wbuf + (phase & 1) * (BN * KP);
#pragma unroll
        for (int kg = 0; kg < KC / 8; ++kg) {
            float4 af[MT], bf[NTL];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const float4*>(P + abase[mt] + kg * 8);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) bf[nt] = *reinterpret_cast<const float4*>(Wb + bbase[nt] + kg * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].x, bf[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].y, bf[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].z, bf[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].w, bf[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }

        if (has_next) store_w(wbuf + ((phase + 1) & 1) * (BN * KP));
        if (last_tap && has_next) store_patch(pbuf + ((chunk + 1) & 1) * patchFloats, chunk + 1);
        __syncthreads();
        if (last_tap) { tap = 0; ++chunk; } else { ++tap; }
    }

    if constexpr (STATS)
        conv_epilogue_stats<BM, BN, WM, WN>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane, smem);
    else
        conv_epilogue<BM, BN, WM, WN, SPLIT>(a, acc, tile_x, tile_y, img0, cout0, wm, wn, lane);
}

// ---- 3x3 convolution with a handful of output channels (the UNet head: 128 -> 3, openaimodel.py:687-691) -----------------
// The MFMA kernel pads Cout to a 128-wide tile: at Cout = 3 97 % of the matrix work is wasted (2.3 ms at 256x256, batch 16,
// 3 TFLOP/s).  Here one thread owns one output pixel and all CO (<= 8) output channels; the 18 x 18 halo patch of a 16 x 16
// pixel tile is staged in LDS per 16-channel chunk (fused GroupNorm -> SiLU coefficients applied while staging, as in the
// MFMA kernel), the [9][CO][16] weights of the chunk come through the scalar cache (wave-uniform addresses).  FMA on the vector ALU: 2 * 9 * Cin * CO FLOP per
// pixel is small next to the 4 * Cin bytes the pixel reads, so the kernel is HBM / LDS bound, not ALU bound.
template <int CO, bool PRE>
__global__ void __launch_bounds__(256) conv3x3_narrow_kernel(const ConvArgs a) {
    constexpr int TS = 16, PS = TS + 2, NPP = PS * PS;             // tile side, patch side, patch pixels
    __shared__ __attribute__((aligned(16))) float patch2[2][NPP * KP];   // double-buffered [patch pixel][16 + 4 pad]
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tilesX = (a.W + TS - 1) / TS;
    const int tile_y = blockIdx.x / tilesX, tile_x = blockIdx.x - tile_y * tilesX;
    const int n = blockIdx.y;
    const int h0 = tile_y * TS - 1, w0 = tile_x * TS - 1;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    const size_t wChunk = (size_t)a.CoutPad * KC;                    // packed weights: [tap][chunk][CoutPad][16]
    // register prefetch: the next chunk's patch is requested before this chunk's FMAs and lands in LDS after them (a single-buffered
    // load -> barrier -> compute -> barrier loop left the loads of a workgroup exposed); SiLU of the fused producer on v_exp / v_rcp
    // (silu_fast, as the Winograd input transform: the IEEE sequence cost as many VALU instructions as the chunk's 576 FMAs)
    constexpr int SLOTS = (NPP * 4 + 255) / 256;
    float4 xr[SLOTS];
    auto request = [&](int chunk) {
        const int cbase = chunk * KC;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            const int pp = f >> 2, c4 = f & 3;
            const int py = pp / PS, px = pp - py * PS;
            const int h = h0 + py, w = w0 + px, c = cbase + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < NPP * 4 && h >= 0 && h < a.H && w >= 0 && w < a.W && c < a.Cin)
                v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + h) * a.W + w) * a.ldx + c);
            xr[s] = v;
        }
    };
    auto land = [&](int chunk) {
        const int cbase = chunk * KC;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            if (f >= NPP * 4) continue;
            const int pp = f >> 2, c4 = f & 3;
            const int py = pp / PS, px = pp - py * PS;
            const int h = h0 + py, w = w0 + px, c = cbase + c4 * 4;
            float4 v = xr[s];
            if (PRE && h >= 0 && h < a.H && w >= 0 && w < a.W && c < a.Cin) {      // (the zero padding stays zero)
                const float4 sc = *reinterpret_cast<const float4*>(a.pre_sc + (size_t)n * a.pre_ld + c);
                const float4 bi = *reinterpret_cast<const float4*>(a.pre_bi + (size_t)n * a.pre_ld + c);
                v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
                if (a.pre_silu) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
            }
            *reinterpret_cast<float4*>(patch2[chunk & 1] + pp * KP + c4 * 4) = v;
        }
    };
    request(0);
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        land(chunk);
        const float* patch = patch2[chunk & 1];
        __syncthreads();                                             // one barrier per chunk: the other patch buffer is free again
        if (chunk + 1 < a.nchunks) request(chunk + 1);
        // the [tap][co][16] weights of this chunk are the same for every thread: read straight from the packed tensor with
        // wave-uniform addresses (scalar loads, FMAs with an SGPR operand) -- no LDS slab, no second barrier
        const float* __restrict__ wq = a.w + (size_t)chunk * wChunk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* P = patch + ((ty + tap / 3) * PS + tx + tap % 3) * KP;
            const float* __restrict__ Wt = wq + (size_t)tap * a.nchunks * wChunk;      // [co (CoutPad rows, zero past Cout)][16]
#pragma unroll
            for (int k4 = 0; k4 < KC; k4 += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(P + k4);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < CO; ++c) acc[c] = fmaf(xs[e], Wt[c * KC + k4 + e], acc[c]);
            }
        }
    }
    const int h = tile_y * TS + ty, w = tile_x * TS + tx;
    if (h < a.H && w < a.W) {
        const size_t pix = (size_t)(n * a.H + h) * a.W + w;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            if (c >= a.Cout) continue;
            const float v = acc[c] + (a.bias ? a.bias[c] : 0.f);
            if (a.out_nchw & 1) a.out[((size_t)(n * a.Cout + c) * a.H + h) * a.W + w] = v;
            else a.out[pix * a.ldo + c] = v;
        }
    }
}

// CONTRACT of this kernel and of the few-output-channel kernels above: the channels of x between the true and the padded input width must
// be ZERO (not merely multiplied by zero weights: 0 x NaN propagates).  bbdm_nchw_to_nhwc_f32 writes them as zeros; every other producer
// of these kernels' inputs writes whole 4-channel groups of real data.
// ---- 3x3 convolution with 4 or 8 (padded) input channels and 128 output channels (the UNet stem: concat(x_t, y) -> 128, ------
// openaimodel.py:524; the latent UNets' 3 -> 128; the VQGAN encoder's first layer) on the f32 MFMA with K = 9 taps x CIN in ONE stage.  The implicit-GEMM kernel above walks K in
// 16-channel chunks per tap -- half of every chunk is padding here, and nine staging rounds with a barrier each carry 8 k of work:
// 0.72 ms at 256x256 / batch 16 for 19 GFLOP and a 537 MB result.  Here a persistent workgroup keeps the [72][128] weights in LDS,
// stages the 18 x 18 x 8 halo patch of a 16 x 16 pixel tile channel-planar (the A fragment of k = (tap, ci), (tap, ci + 1) is two
// conflict-free rows of one plane each, all offsets immediates), prefetches the next tile's patch into registers, and each wave
// multiplies 64 pixels x 128 channels: 288 MFMAs per tile.  STATS: GroupNorm statistics of the output as conv_epilogue_stats.
template <int CIN, bool STATS>
__global__ void __launch_bounds__(256, 2) conv3x3_stem_kernel(const ConvArgs a, int tiles_total) {
    constexpr int TS = 16, PR = 18, NPP = PR * PR, PLANE = 328, WP = 160, Q = CIN / 4, SLOTS = (NPP * Q + 255) / 256;
    __shared__ float patch[CIN * PLANE];                         // [ci][py * 18 + px]
    __shared__ float wsm[9 * CIN * WP];                          // [k = tap * CIN + ci][co], pitch = 32 (mod 64): k and k + 1 on disjoint banks
    __shared__ unsigned long long ls[128 * SA_W];                // [2 consumers][32 groups][2] exact limb accumulators (stats_acc.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 9 * CIN * 128; i += 256) {
        const int k = i >> 7, co = i & 127, tap = k / CIN, ci = k % CIN;
        wsm[k * WP + co] = a.w[((size_t)tap * a.nchunks * a.CoutPad + co) * KC + ci];
    }
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = a.bias ? a.bias[cb * 32 + l31] : 0.f;
    const int tilesX = (a.W + TS - 1) / TS, tilesY = (a.H + TS - 1) / TS, per_img = tilesX * tilesY;
    float4 xr[SLOTS];
    auto request = [&](int t) {
        const int n = t / per_img, rem = t - n * per_img, ty = rem / tilesX, tx = rem - ty * tilesX;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = tid + s * 256, pix = i / Q, half = i % Q;
            const int py = pix / PR, px = pix - py * PR;
            const int h = ty * TS + py - 1, w = tx * TS + px - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < NPP * Q && h >= 0 && h < a.H && w >= 0 && w < a.W)
                v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + h) * a.W + w) * a.ldx + half * 4);
            xr[s] = v;
        }
    };
    auto land = [&]() {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = tid + s * 256, pix = i / Q, half = i % Q;
            if (i < NPP * Q) {
                float* d = patch + (half * 4) * PLANE + pix;
                d[0] = xr[s].x; d[PLANE] = xr[s].y; d[2 * PLANE] = xr[s].z; d[3 * PLANE] = xr[s].w;
            }
        }
    };
    if (STATS) for (int i = tid; i < 128 * SA_W; i += 256) ls[i] = 0ull;      // (the first barrier of the tile loop orders it)
    int t = blockIdx.x;
    if (t < tiles_total) request(t);
    const float* ab = patch + hi * PLANE + (wave * 4 + (l31 >> 4)) * PR + (l31 & 15);
    const float* bb = wsm + hi * WP + l31;
    for (; t < tiles_total; t += gridDim.x) {
        land();
        __syncthreads();
        if (t + (int)gridDim.x < tiles_total) request(t + gridDim.x);
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int cp = 0; cp < CIN / 2; ++cp) {
                const int ko = (2 * cp) * PLANE + (tap / 3) * PR + tap % 3;
                const float a0 = ab[ko], a1 = ab[ko + 2 * PR];
                const float* bk = bb + (tap * CIN + 2 * cp) * WP;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const float b = bk[cb * 32];
                    acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][cb], 0, 0, 0);
                    acc[1][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][cb], 0, 0, 0);
                }
            }
        const int n = t / per_img, rem = t - n * per_img, ty = rem / tilesX, tx = rem - ty * tilesX;
        double psum[4] = {0.0, 0.0, 0.0, 0.0}, psq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int h = ty * TS + wave * 4 + i * 2 + (m >> 4), w = tx * TS + (m & 15);
                if (h < a.H && w < a.W) {
                    float* o = a.out + ((size_t)(n * a.H + h) * a.W + w) * a.ldo + l31;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const float v = acc[i][cb][r] + bv[cb];
                        o[cb * 32] = v;
                        if (STATS) { psum[cb] += (double)v; psq[cb] += (double)v * v; }
                    }
                }
            }
        if (STATS) {
            // as conv_epilogue_stats: the two halves of the wave (same channels, other pixels), then the lanes of a group by a
            // segmented scan, then ONE exact LDS add per group and wave
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int co = cb * 32 + l31;
                const double hs = psum[cb] + __shfl_xor(psum[cb], 32), hq = psq[cb] + __shfl_xor(psq[cb], 32);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (!a.st_s[k]) continue;
                    const int g = (a.st_coff[k] + co) / a.st_cpg[k];
                    double vs = hs, vq = hq;
                    sa_seg_scan2(vs, vq, g, lane, a.st_cpg[k] < 32 ? a.st_cpg[k] : 32, 32);
                    const bool tail = sa_seg_tail(g, lane, 32);
                    if (lane < 32 && tail) {
                        sa_add(ls + (size_t)((k * 32 + g) * 2) * SA_W, vs);
                        sa_add(ls + (size_t)((k * 32 + g) * 2 + 1) * SA_W, vq);
                    }
                }
            }
        }
        __syncthreads();                                         // patch readers done (and the tile's statistics complete)
        if (STATS && tid < 128) {
            const int k = tid >> 6;
            if (a.st_s[k]) sa_add_cell(a.st_s[k] + ((size_t)n * 64 + (tid & 63)) * SA_W, ls + (size_t)tid * SA_W);
#pragma unroll
            for (int i = 0; i < SA_W; ++i) ls[(size_t)tid * SA_W + i] = 0ull;      // ready for the next tile (the same thread reads and clears)
        }
    }
}

// The same for small images (below 65536 output pixels: the heads of the latent UNets and of the 64x64 pixel model): 16 x 16 tiles
// leave 32 - 64 workgroups on 256 CUs, each walking all channel chunks alone (0.12 ms for the 8192 pixels of the LBBDM-f16 head).
// Here a workgroup owns an 8 x 8 tile and its four waves split the chunks (wave w: chunks w, w + 4, ...; its own double-buffered
// 10 x 10 patch), the four partial sums meet in LDS in a fixed order: 16 x the parallelism.
template <int CO, bool PRE>
__global__ void __launch_bounds__(256) conv3x3_narrow_small_kernel(const ConvArgs a) {
    constexpr int TS = 8, PS = TS + 2, NPP = PS * PS, SLOTS = (NPP * 4 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float patchw[4][2][NPP * KP];   // [wave][buffer][patch pixel][16 + 4 pad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = lane & 7, ty = lane >> 3;
    const int tilesX = (a.W + TS - 1) / TS;
    const int tile_y = blockIdx.x / tilesX, tile_x = blockIdx.x - tile_y * tilesX;
    const int n = blockIdx.y;
    const int h0 = tile_y * TS - 1, w0 = tile_x * TS - 1;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    const size_t wChunk = (size_t)a.CoutPad * KC;
    float4 xr[SLOTS];
    auto request = [&](int chunk) {
        const int cbase = chunk * KC;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = lane + s * 64;
            const int pp = f >> 2, c4 = f & 3;
            const int py = pp / PS, px = pp - py * PS;
            const int h = h0 + py, w = w0 + px, c = cbase + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (chunk < a.nchunks && f < NPP * 4 && h >= 0 && h < a.H && w >= 0 && w < a.W && c < a.Cin)
                v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + h) * a.W + w) * a.ldx + c);
            xr[s] = v;
        }
    };
    auto land = [&](int chunk, float* patch) {
        const int cbase = chunk * KC;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = lane + s * 64;
            if (f >= NPP * 4) continue;
            const int pp = f >> 2, c4 = f & 3;
            const int py = pp / PS, px = pp - py * PS;
            const int h = h0 + py, w = w0 + px, c = cbase + c4 * 4;
            float4 v = xr[s];
            if (PRE && chunk < a.nchunks && h >= 0 && h < a.H && w >= 0 && w < a.W && c < a.Cin) {
                const float4 sc = *reinterpret_cast<const float4*>(a.pre_sc + (size_t)n * a.pre_ld + c);
                const float4 bi = *reinterpret_cast<const float4*>(a.pre_bi + (size_t)n * a.pre_ld + c);
                v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
                if (a.pre_silu) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
            }
            *reinterpret_cast<float4*>(patch + pp * KP + c4 * 4) = v;
        }
    };
    const int rounds = (a.nchunks + 3) / 4;                       // every wave runs every round (the barriers are workgroup-wide)
    request(wave);
    for (int r = 0; r < rounds; ++r) {
        const int chunk = r * 4 + wave;
        float* patch = patchw[wave][r & 1];
        land(chunk, patch);
        __syncthreads();
        if (r + 1 < rounds) request(chunk + 4);
        if (chunk < a.nchunks) {
            const float* __restrict__ wq = a.w + (size_t)chunk * wChunk;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* P = patch + ((ty + tap / 3) * PS + tx + tap % 3) * KP;
                const float* __restrict__ Wt = wq + (size_t)tap * a.nchunks * wChunk;
#pragma unroll
                for (int k4 = 0; k4 < KC; k4 += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(P + k4);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < CO; ++c) acc[c] = fmaf(xs[e], Wt[c * KC + k4 + e], acc[c]);
                }
            }
        }
    }
    __syncthreads();                                              // the patches are free: the partial sums meet in them
    float* red = &patchw[0][0][0];                                // [4 waves][64 pixels][CO]
#pragma unroll
    for (int c = 0; c < CO; ++c) red[(wave * 64 + lane) * CO + c] = acc[c];
    __syncthreads();
    const int h = tile_y * TS + ty, w = tile_x * TS + tx;
    if (wave == 0 && h < a.H && w < a.W) {
        const size_t pix = (size_t)(n * a.H + h) * a.W + w;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            if (c >= a.Cout) continue;
            const float v = ((red[lane * CO + c] + red[(64 + lane) * CO + c]) + (red[(128 + lane) * CO + c] + red[(192 + lane) * CO + c])) +
                            (a.bias ? a.bias[c] : 0.f);
            if (a.out_nchw & 1) a.out[((size_t)(n * a.Cout + c) * a.H + h) * a.W + w] = v;
            else a.out[pix * a.ldo + c] = v;
        }
    }
}

template <int CO>
void launch_narrow(const ConvArgs& a, hipStream_t st) {
    if ((long long)a.N * a.H * a.W < 65536) {                    // small images: 8 x 8 tiles, the chunks split over the four waves
        const dim3 grid8((unsigned)(cdiv(a.W, 8) * cdiv(a.H, 8)), (unsigned)a.N);
        if (a.pre_sc) hipLaunchKernelGGL((conv3x3_narrow_small_kernel<CO, true>), grid8, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_narrow_small_kernel<CO, false>), grid8, dim3(256), 0, st, a);
        return;
    }
    const dim3 grid((unsigned)(cdiv(a.W, 16) * cdiv(a.H, 16)), (unsigned)a.N);
    if (a.pre_sc) hipLaunchKernelGGL((conv3x3_narrow_kernel<CO, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_narrow_kernel<CO, false>), grid, dim3(256), 0, st, a);
}

// out = sum_s ws[s] + bias (+ residual): fixed summation order -> deterministic
__global__ void conv_splitk_reduce_kernel(const ConvArgs a) {
    const size_t M = (size_t)a.N * a.H * a.W;
    const size_t total = M * a.Cout;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % a.Cout);
        const size_t pix = i / a.Cout;
        float v = 0.f;
        for (int s = 0; s < a.splits; ++s) v += a.ws[((size_t)s * M + pix) * a.ldw + co];
        if (a.bias) v += a.bias[co];
        const size_t n = pix / ((size_t)a.H * a.W);
        if (a.res) v += (a.out_nchw & 2) ? a.res[n * a.ldr + co] : a.res[pix * a.ldr + co];
        if (a.out_nchw & 1) {
            const size_t hw = pix - n * (size_t)a.H * a.W;
            a.out[(n * a.Cout + co) * (size_t)a.H * a.W + hw] = v;
        } else {
            a.out[pix * a.ldo + co] = v;
        }
    }
}

template <typename K>
int set_lds_limit(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes) == hipSuccess
               ? 0
               : -1;
}

template <int BM, int BN, int WM, int WN, int PSLOTS, int OCC, int VAR>
int launch_variant(const ConvArgs& a, size_t lds, long long blocks, hipStream_t stream) {
    auto kern = conv_igemm_f32<BM, BN, WM, WN, PSLOTS, OCC, VAR>;
    static size_t lds_set_dev[BBDM_MAX_DEVICES] = {};
    size_t& lds_set = lds_set_dev[bbdm_device_slot()];
    if (lds > lds_set) {
        if (set_lds_limit(kern, lds) != 0) {
            bbdm_set_error("conv: hipFuncSetAttribute(%zu B LDS) failed", lds);
            return BBDM_E_LAUNCH;
        }
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, a.splits, a.batch), dim3(WM * WN * 64), lds, stream, a);
    return 0;
}

template <int BM, int BN, int WM, int WN, int PSLOTS, int OCC>
int launch_conv(ConvArgs& a, hipStream_t stream) {
    const int TWc = ceil_pow2(a.W) < 32 ? ceil_pow2(a.W) : 32;
    int THc = BM / TWc;
    if (THc > ceil_pow2(a.H)) THc = ceil_pow2(a.H);
    int IM = BM / (TWc * THc);
    a.TWl = ilog2(TWc);
    a.THl = ilog2(THc);
    a.PW = TWc + 2 * a.pad;
    a.PH = THc + 2 * a.pad;
    // tiny images: cap the images per block so that the halo patch fits the staging slots (the remaining rows of
    // the M tile are padding)
    int maxIm = (PSLOTS * WM * WN * 64 / (KC / 4)) / (a.PW * a.PH);
    if (maxIm > 32) maxIm = 32;          // the per-slot image index is kept in 5 bits
    if (IM > maxIm) IM = maxIm;
    if (IM > a.N) IM = a.N;
    if (IM < 1) return 1;
    a.imgs = IM;
    a.patchPix = IM * a.PW * a.PH;
    a.tilesX = cdiv(a.W, TWc);
    a.tilesY = cdiv(a.H, THc);
    a.tilesN = cdiv(a.Cout, BN);
    if (a.patchPix * (KC / 4) > PSLOTS * WM * WN * 64) return 1;   // does not fit this instantiation
    const size_t lds = ((size_t)2 * a.patchPix * KP + 2 * BN * KP) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    const long long blocks = (long long)a.tilesN * a.tilesX * a.tilesY * cdiv(a.N, IM);
    // split-K (a.splits chosen by conv_plan): spread the Cin chunks over gridDim.y, bounded by the workspace
    {
        int want = a.splits;
        a.splits = 1;
        a.chunks_per_split = a.nchunks;
        if (a.ws && want > 1) {
            const size_t per = (size_t)a.N * a.H * a.W * a.ldw;
            if ((size_t)want * per > a.ws_cap) want = (int)(a.ws_cap / per);
            if (want > 1) {
                a.chunks_per_split = cdiv(a.nchunks, want);
                a.splits = cdiv(a.nchunks, a.chunks_per_split);
            }
        }
    }
    const int var = (a.pre_sc ? 1 : 0) | (a.splits > 1 ? 2 : 0);
    int lrc = 0;
    bool tile_gemm = false;
    if constexpr (BM == 256 && PSLOTS == 2)          // the batched tile-GEMM instantiation (bbdm_conv1x1_batched)
        tile_gemm = var == 0 && a.taps == 1;
    if (tile_gemm) {
        if constexpr (BM == 256 && PSLOTS == 2) lrc = launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 4>(a, lds, blocks, stream);
    } else if (a.st_s[0] || a.st_s[1]) {
        if (a.splits > 1 || IM != 1 || (a.out_nchw & 1)) return 2;          // not fusable: the caller asked for the impossible
        lrc = a.pre_sc ? launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 9>(a, lds, blocks, stream)
                       : launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 8>(a, lds, blocks, stream);
    } else {
        switch (var) {
            case 0: lrc = launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 0>(a, lds, blocks, stream); break;
            case 1: lrc = launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 1>(a, lds, blocks, stream); break;
            case 2: lrc = launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 2>(a, lds, blocks, stream); break;
            default: lrc = launch_variant<BM, BN, WM, WN, PSLOTS, OCC, 3>(a, lds, blocks, stream); break;
        }
    }
    if (lrc != 0) return lrc;
    if (a.splits > 1) {
        const size_t total = (size_t)a.N * a.H * a.W * a.Cout;
        size_t rb = (total + 255) / 256;
        if (rb > 4096) rb = 4096;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, a);
    }
    return 0;
}

}  // namespace

extern "C" size_t bbdm_conv_packed_floats(int Cout, int CinPad, int ks) {
    const int CoutPad = cdiv(Cout, 128) * 128;
    const int nchunks = cdiv(CinPad, KC);
    return (size_t)ks * ks * nchunks * CoutPad * KC;
}

extern "C" int bbdm_conv_pack_weight_f32(const float* w_oihw, float* packed, int Cout, int Cin, int CinPad, int ks,
                                         void* stream) {
    BBDM_REQUIRE(w_oihw && packed, "conv_pack: null pointer");
    BBDM_REQUIRE((ks == 1 || ks == 3) && Cout > 0 && Cin > 0 && CinPad >= Cin && CinPad % 4 == 0,
                 "conv_pack: bad shape Cout=%d Cin=%d CinPad=%d ks=%d", Cout, Cin, CinPad, ks);
    const int CoutPad = cdiv(Cout, 128) * 128;
    const int nchunks = cdiv(CinPad, KC);
    const size_t total = bbdm_conv_packed_floats(Cout, CinPad, ks);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout, Cin,
                       CinPad, ks, CoutPad, nchunks);
    BBDM_CHECK_LAUNCH("conv_pack");
    return BBDM_OK;
}

// Batched activation-by-activation GEMM (VQGAN AttnBlock, model/VQGAN/model.py:166-185: w_ = bmm(q, k), h_ = bmm(v, w_)): the
// B operand of batch element b is a ROW-MAJOR activation matrix Bm_b [R x K] (transposed == 0: out = A Bm^T, e.g. q k^T) or
// [K x R] (transposed != 0: out = A Bm, e.g. softmax(w) v), packed into the kernel's weight layout by this kernel.
__global__ void pack_activation_kernel(const float* __restrict__ src, size_t sz, int lds_, float* __restrict__ p, size_t pz, int R,
                                       int K, int RPad, int nchunks, int transposed) {
    const size_t per = (size_t)nchunks * RPad * KC;
    const float* s = src + (size_t)blockIdx.y * sz;
    float* d = p + (size_t)blockIdx.y * pz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        const int k = i % KC;
        size_t t = i / KC;
        const int r = t % RPad;
        const int chunk = t / RPad;
        const int kk = chunk * KC + k;
        float v = 0.0f;
        if (r < R && kk < K) v = transposed ? s[(size_t)kk * lds_ + r] : s[(size_t)r * lds_ + kk];
        d[i] = v;
    }
}

// Tile / split-K choice.  The 256x128 tile (8 waves, 2 blocks per CU) is the efficient one; it needs >= 512 blocks to
// fill the chip, so problems with fewer output tiles get the Cin reduction split over 2..32 extra workgroups each
// (partials summed in a fixed order by conv_splitk_reduce_kernel).
struct ConvPlan {
    bool big;
    int splits;
};
static ConvPlan conv_plan(long long M, int Cout, int nchunks) {
    ConvPlan p;
    const long long b256 = ((M + 255) / 256) * cdiv(Cout, 128);
    const long long b128 = ((M + 127) / 128) * cdiv(Cout, 128);
    if (b256 >= 512) { p.big = true; p.splits = 1; return p; }
    if (b256 >= 128 && nchunks >= 8) {            // 2..4-way split of the big tile
        p.big = true;
        p.splits = (int)cdiv(512, (int)b256);
        return p;
    }
    p.big = false;
    p.splits = 1;
    if (b128 < 256 && nchunks >= 4) {
        // as many splits as keep every workgroup resident at once (two 8-wave workgroups per CU = 512): rounding UP gave the qkv
        // projections of the latent models (512 x 1024 -> 3072: 96 tiles x 6 = 576 workgroups) a second, nearly empty round
        int want = 512 / (int)b128;
        if (want > nchunks / 2) want = nchunks / 2;
        if (want > 32) want = 32;
        p.splits = want > 1 ? want : 1;
    }
    return p;
}

extern "C" size_t bbdm_conv_splitk_workspace_floats(int N, int H, int W, int CinPad, int Cout, int ks) {
    (void)ks;
    const long long M = (long long)N * H * W;
    const ConvPlan p = conv_plan(M, Cout, cdiv(CinPad, KC));
    return p.splits > 1 ? (size_t)p.splits * M * ((Cout + 3) & ~3) : 0;
}

// Batched 1x1 "convolution" = `batch` independent GEMMs [pixels x CinPad] x [CinPad x Cout] in one launch (used by the
// Winograd path; declared in common.h).  x / packed_w / out advance by xz / wz / oz floats per batch element.
int bbdm_conv1x1_batched(const float* x, int ldx, size_t xz, const float* packed_w, size_t wz, float* out, int ldo, size_t oz,
                         int batch, int H, int W, int CinPad, int Cout, hipStream_t st) {
    ConvArgs a;
    a.x = x; a.w = packed_w; a.bias = nullptr; a.res = nullptr; a.out = out;
    a.ldx = ldx; a.ldr = 0; a.ldo = ldo; a.out_nchw = 0;
    a.N = 1; a.H = H; a.W = W; a.Cin = CinPad; a.Cout = Cout;
    a.CoutPad = cdiv(Cout, 128) * 128;
    a.taps = 1; a.nchunks = cdiv(CinPad, KC); a.pad = 0;
    a.ws = nullptr; a.ws_cap = 0; a.ldw = (Cout + 3) & ~3;
    a.pre_sc = nullptr; a.pre_bi = nullptr; a.pre_ld = 0; a.pre_silu = 0;
    a.batch = batch; a.xz = xz; a.wz = wz; a.oz = oz;
    a.st_s[0] = a.st_s[1] = nullptr; a.st_cpg[0] = a.st_cpg[1] = 1; a.st_coff[0] = a.st_coff[1] = 0;
    a.splits = 1;
    int rc = launch_conv<256, 128, 4, 2, 2, 2>(a, st);
    if (rc == 1) {
        bbdm_set_error("conv1x1_batched: tile configuration does not fit H=%d W=%d", H, W);
        return BBDM_E_BADARG;
    }
    return rc;
}

// Can bbdm_conv2d_nhwc_stats_f32 accumulate GroupNorm statistics for this shape?  (no split-K, one image per workgroup tile,
// not the narrow / NCHW head)
extern "C" int bbdm_conv_stats_fusable(int N, int H, int W, int CinPad, int Cout, int ks) {
    if (Cout % 4 || Cout <= 8) return 0;
    const long long M = (long long)N * H * W;
    const ConvPlan p = conv_plan(M, Cout, cdiv(CinPad, KC));
    if (p.splits > 1) return 0;
    const int BMv = p.big ? 256 : 128;
    const int TWc = ceil_pow2(W) < 32 ? ceil_pow2(W) : 32;
    int THc = BMv / TWc;
    if (THc > ceil_pow2(H)) THc = ceil_pow2(H);
    return TWc * THc == BMv ? 1 : 0;
}

extern "C" int bbdm_conv2d_nhwc_stats_f32(const float* x, int ldx, const float* packed_w, const float* bias,
                                          const float* residual, int ldr, float* out, int ldo, int out_nchw, float* ws,
                                          size_t ws_floats, const float* pre_scale, const float* pre_bias, int pre_ld,
                                          int pre_silu, int N, int H, int W, int CinPad, int Cout, int ks, void* stats0,
                                          int cpg0, int coff0, void* stats1, int cpg1, int coff1, void* stream) {
    BBDM_REQUIRE(x && packed_w && out, "conv2d: null pointer");
    BBDM_REQUIRE(ks == 1 || ks == 3, "conv2d: ks=%d unsupported (1 or 3)", ks);
    BBDM_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && CinPad > 0, "conv2d: bad shape");
    BBDM_REQUIRE(CinPad % 4 == 0 && ldx % 4 == 0 && ldx >= CinPad, "conv2d: CinPad=%d ldx=%d must be multiples of 4",
                 CinPad, ldx);
    BBDM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "conv2d: x / w must be 16-byte aligned");
    BBDM_REQUIRE((size_t)N * H * W * (size_t)ldx < (1ull << 32), "conv2d: input exceeds 2^32 elements");
    BBDM_REQUIRE((out_nchw & 1) || ldo >= Cout, "conv2d: ldo < Cout");
    BBDM_REQUIRE((out_nchw & ~3) == 0, "conv2d: unknown flag bits 0x%x", out_nchw);
    BBDM_REQUIRE(!residual || ldr >= Cout, "conv2d: ldr < Cout");
    ConvArgs a;
    a.x = x; a.w = packed_w; a.bias = bias; a.res = residual; a.out = out;
    a.ldx = ldx; a.ldr = ldr; a.ldo = ldo; a.out_nchw = out_nchw;
    a.N = N; a.H = H; a.W = W; a.Cin = CinPad; a.Cout = Cout;
    a.CoutPad = cdiv(Cout, 128) * 128;
    a.taps = ks * ks; a.nchunks = cdiv(CinPad, KC); a.pad = ks / 2;
    a.ws = ws; a.ws_cap = ws ? ws_floats : 0; a.ldw = (Cout + 3) & ~3;
    BBDM_REQUIRE((pre_scale == nullptr) == (pre_bias == nullptr), "conv2d: pre_scale / pre_bias must come together");
    BBDM_REQUIRE(!pre_scale || (pre_ld % 4 == 0 && pre_ld >= CinPad && (((uintptr_t)pre_scale | (uintptr_t)pre_bias) & 15) == 0),
                 "conv2d: pre_ld / alignment of the fused-producer coefficients");
    a.pre_sc = pre_scale; a.pre_bi = pre_bias; a.pre_ld = pre_ld; a.pre_silu = pre_silu;
    a.batch = 1; a.xz = a.wz = a.oz = 0;
    BBDM_REQUIRE((!stats0 && !stats1) || bbdm_conv_stats_fusable(N, H, W, CinPad, Cout, ks),
                 "conv2d: GroupNorm statistics cannot be accumulated for this shape (bbdm_conv_stats_fusable)");
    BBDM_REQUIRE((!stats0 || (cpg0 > 0 && coff0 >= 0 && (coff0 + Cout - 1) / cpg0 < 32)) &&
                     (!stats1 || (cpg1 > 0 && coff1 >= 0 && (coff1 + Cout - 1) / cpg1 < 32)),
                 "conv2d: statistics targets need (coff + Cout) / cpg <= 32");
    a.st_s[0] = (unsigned long long*)stats0; a.st_cpg[0] = cpg0 > 0 ? cpg0 : 1; a.st_coff[0] = coff0;
    a.st_s[1] = (unsigned long long*)stats1; a.st_cpg[1] = cpg1 > 0 ? cpg1 : 1; a.st_coff[1] = coff1;
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)N * H * W;
    const long long narrow_min = 4096;       // (below: the general implicit GEMM; measured slower from 8192 pixels)
    if (ks == 3 && Cout <= 8 && !residual && M >= narrow_min) {      // a few output channels: one thread per pixel (see above)
        if (Cout <= 3) launch_narrow<3>(a, st); else if (Cout <= 4) launch_narrow<4>(a, st); else launch_narrow<8>(a, st);
        BBDM_CHECK_LAUNCH("conv2d(narrow)");
        return BBDM_OK;
    }
    if (ks == 3 && (CinPad == 8 || CinPad == 4) && Cout == 128 && !residual && !pre_scale && out_nchw == 0 && M >= 4096) {
        const int tiles = N * cdiv(H, 16) * cdiv(W, 16);         // the stem: K = 72 / 36 in one stage (conv3x3_stem_kernel)
        const dim3 grid((unsigned)(tiles < 512 ? tiles : 512));
        const bool stats = stats0 || stats1;
        if (CinPad == 8) {
            if (stats) hipLaunchKernelGGL((conv3x3_stem_kernel<8, true>), grid, dim3(256), 0, st, a, tiles);
            else hipLaunchKernelGGL((conv3x3_stem_kernel<8, false>), grid, dim3(256), 0, st, a, tiles);
        } else {
            if (stats) hipLaunchKernelGGL((conv3x3_stem_kernel<4, true>), grid, dim3(256), 0, st, a, tiles);
            else hipLaunchKernelGGL((conv3x3_stem_kernel<4, false>), grid, dim3(256), 0, st, a, tiles);
        }
        BBDM_CHECK_LAUNCH("conv2d(stem)");
        return BBDM_OK;
    }
    const ConvPlan plan = conv_plan(M, Cout, a.nchunks);
    a.splits = plan.splits;
    int rc;
    if (plan.big) {
        rc = launch_conv<256, 128, 4, 2, 3, 2>(a, st);
        if (rc == 1) { a.splits = plan.splits; rc = launch_conv<256, 128, 4, 2, 5, 2>(a, st); }
    } else {
        rc = launch_conv<128, 128, 4, 2, 2, 2>(a, st);            // 8 waves of 32x64: 16 waves per CU like the big tile
        if (rc == 1) { a.splits = plan.splits; rc = launch_conv<128, 128, 4, 2, 3, 2>(a, st); }
        if (rc == 1) { a.splits = plan.splits; rc = launch_conv<128, 128, 2, 2, 6, 2>(a, st); }
    }
    if (rc == 1 || rc == 2) {
        bbdm_set_error(rc == 1 ? "conv2d: no tile configuration fits N=%d H=%d W=%d"
                               : "conv2d: statistics requested for a launch that cannot accumulate them (N=%d H=%d W=%d)", N, H, W);
        return BBDM_E_BADARG;
    }
    if (rc < 0) return rc;
    BBDM_CHECK_LAUNCH("conv2d");
    return BBDM_OK;
}

extern "C" int bbdm_conv2d_nhwc_f32(const float* x, int ldx, const float* packed_w, const float* bias,
                                    const float* residual, int ldr, float* out, int ldo, int out_nchw, float* ws,
                                    size_t ws_floats, const float* pre_scale, const float* pre_bias, int pre_ld,
                                    int pre_silu, int N, int H, int W, int CinPad, int Cout, int ks, void* stream) {
    return bbdm_conv2d_nhwc_stats_f32(x, ldx, packed_w, bias, residual, ldr, out, ldo, out_nchw, ws, ws_floats, pre_scale, pre_bias,
                                      pre_ld, pre_silu, N, H, W, CinPad, Cout, ks, nullptr, 0, 0, nullptr, 0, 0, stream);
}

// ---- batched GEMM with activation operands (see pack_activation_kernel) ------------------------------------------------
extern "C" size_t bbdm_gemm_packed_b_floats(int R, int K) { return (size_t)cdiv(K, KC) * (cdiv(R, 128) * 128) * KC; }

extern "C" int bbdm_gemm_pack_b_f32(const float* b, int ldb, size_t b_stride, float* packed, int batch, int R, int K,
                                    int transposed, void* stream) {
    BBDM_REQUIRE(b && packed && batch > 0 && R > 0 && K > 0 && ldb >= (transposed ? R : K), "gemm_pack_b: bad args");
    const size_t per = bbdm_gemm_packed_b_floats(R, K);
    unsigned blocks = (unsigned)((per + 255) / 256 > 2048 ? 2048 : (per + 255) / 256);
    hipLaunchKernelGGL(pack_activation_kernel, dim3(blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, b, b_stride, ldb,
                       packed, per, R, K, cdiv(R, 128) * 128, cdiv(K, KC), transposed);
    BBDM_CHECK_LAUNCH("gemm_pack_b");
    return BBDM_OK;
}

extern "C" int bbdm_gemm_batched_f32(const float* a, int lda, size_t a_stride, const float* packed_b, float* out, int ldo,
                                     size_t out_stride, int batch, int rows, int K, int R, void* stream) {
    // out_b [rows x R] = A_b [rows x K] . B_b   (B_b as packed by bbdm_gemm_pack_b_f32); K % 4 == 0
    BBDM_REQUIRE(a && packed_b && out && batch > 0 && rows > 0 && K > 0 && R > 0, "gemm_batched: bad args");
    BBDM_REQUIRE(K % 4 == 0 && lda % 4 == 0 && lda >= K && ldo >= R && a_stride % 4 == 0 && ((uintptr_t)a & 15) == 0,
                 "gemm_batched: K=%d lda=%d ldo=%d must be multiples of 4 / aligned", K, lda, ldo);
    BBDM_REQUIRE((size_t)rows * (size_t)lda < (1ull << 32), "gemm_batched: one A matrix exceeds 2^32 elements");
    // rows are laid out as an image of width 32 when possible (whole 8x32 spatial tiles), else as one column
    int H = rows, W = 1;
    if (rows % 32 == 0) { H = rows / 32; W = 32; }
    int rc = bbdm_conv1x1_batched(a, lda, a_stride, packed_b, bbdm_gemm_packed_b_floats(R, K), out, ldo, out_stride, batch, H, W, K,
                                  R, (hipStream_t)stream);
    if (rc != BBDM_OK) return rc;
    BBDM_CHECK_LAUNCH("gemm_batched");
    return BBDM_OK;
}
