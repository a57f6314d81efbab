// conv_wgrad.hip -- weight / bias gradients of the fp32 NHWC convolutions (training path) and the weight packer
// for the data-gradient pass.
//
// Reference: autograd of nn.Conv2d / nn.Conv1d(k=1) at openaimodel.py:207,233,244,307,315,524,690 (the reference
// relies on ATen's cudnn/MIOpen backward; here it is explicit).
//
//   dW[co][ci][r][s] = sum_{n,h,w} dY[n,h,w,co] * X[n,h+r-pad,w+s-pad,ci]        (this file, MFMA fp32)
//   db[co]           = sum_{n,h,w} dY[n,h,w,co]                                   (this file, HBM-bound)
//   dX               = conv(dY, W^T flipped)   -> the forward kernel (conv_igemm.hip) with the weights packed by
//                                                bbdm_conv_pack_weight_dgrad_f32 below.
//
// wgrad is a GEMM whose K dimension is the pixel index: C_tap[ci][co] = sum_m Xtap[m][ci] * dY[m][co].  One
// workgroup (4 waves, 2x2) owns a 64(ci) x 64(co) tile for ALL taps (9 accumulators of 32x32 per wave) and walks
// 8x8-pixel tiles: the X halo patch (10x10 pixels) and the dY tile are staged in LDS once per pixel tile and shared
// by the 9 taps, the dY fragment is read once per pixel pair and feeds 9 MFMAs.  The pixel loop is split over
// gridDim.z; partial tiles go to a workspace and a second kernel sums the splits in a fixed order (deterministic)
// while transposing to the parameter's OIHW layout.
#include "common.h"
#include "stats_acc.h"
#include "bf3_split.h"
#include <stdlib.h>

namespace {

constexpr int CT = 64;           // channels per tile (both ci and co)
constexpr int CP = CT + 4;       // LDS pitch
constexpr int PTH = 8, PTW = 8;  // pixel tile

struct WgradArgs {
    const float* x;
    const float* dy;
    float* ws;                   // [splits][taps][CinT*64][CoutT*64] partial tiles (padded dims)
    float* wsb;                  // [splits][CoutP] partial bias gradients (column sums of dY), or nullptr
    int ldx, ldy;
    int N, H, W, Cin, Cout;
    int taps, pad;
    int tilesX, tilesY;          // pixel tiles per image
    int ptiles;                  // N * tilesX * tilesY
    int tiles_per_split;
    int CinP, CoutP;             // Cin rounded up to the ci tile (64 or 128), Cout rounded up to 64
};

template <int TAPS, int WI>
__global__ void __launch_bounds__(WI * 128, 1) conv_wgrad_f32(const WgradArgs a) {
    constexpr int NT = WI * 128;                     // threads: WI ci-waves x 2 co-waves
    constexpr int CTI = WI * 32;                     // ci channels per block
    constexpr int CPI = CTI + 4;                     // LDS pitch of the X patch
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int PW = PTW + 2 * PAD, PH = PTH + 2 * PAD;
    constexpr int XPIX = PW * PH;                    // 100 or 64
    constexpr int XSLOTS = (XPIX * CTI / 4 + NT - 1) / NT;
    constexpr int YSLOTS = (PTH * PTW * CT / 4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][XPIX][CPI] + [2][64][CP]
    float* xbuf = smem;
    float* ybuf = smem + 2 * XPIX * CPI;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wo = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int ci0 = blockIdx.x * CTI, co0 = blockIdx.y * CT;
    const int t_begin = blockIdx.z * a.tiles_per_split;
    const int t_end = min(a.ptiles, t_begin + a.tiles_per_split);

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // bias gradient = column sums of dY: one extra MFMA per pixel pair with an all-ones A operand (every row of the
    // result is the column sum), only in the waves that own ci-rows 0..31 of ci-tile 0 -- no extra memory traffic.
    const bool do_bias = a.wsb != nullptr && blockIdx.x == 0 && wi == 0;
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;

    float4 xreg[XSLOTS], yreg[YSLOTS];
    auto load_tile = [&](int t) {
        const int n = t / (a.tilesX * a.tilesY);
        const int rem = t - n * (a.tilesX * a.tilesY);
        const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        const int h0 = ty * PTH, w0 = tx * PTW;
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = tid + s * NT;
            const int pp = f / (CTI / 4), c = (f % (CTI / 4)) * 4;
            const int py = pp / PW, px = pp - py * PW;
            const int h = h0 + py - PAD, w = w0 + px - PAD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < XPIX * CTI / 4 && h >= 0 && h < a.H && w >= 0 && w < a.W && ci0 + c < a.Cin)
                v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + h) * a.W + w) * a.ldx + ci0 + c);
            xreg[s] = v;
        }
#pragma unroll
        for (int s = 0; s < YSLOTS; ++s) {
            const int f = tid + s * NT;
            const int pp = f / (CT / 4), c = (f % (CT / 4)) * 4;
            const int py = pp / PTW, px = pp - py * PTW;
            const int h = h0 + py, w = w0 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < PTH * PTW * CT / 4 && h < a.H && w < a.W && co0 + c < a.Cout) {
                const float* p = a.dy + ((size_t)(n * a.H + h) * a.W + w) * a.ldy + co0 + c;
                if (co0 + c + 3 < a.Cout) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {                                   // ragged Cout (e.g. the 3-channel head)
                    v.x = p[0];
                    if (co0 + c + 1 < a.Cout) v.y = p[1];
                    if (co0 + c + 2 < a.Cout) v.z = p[2];
                }
            }
            yreg[s] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int f = tid + s * NT;
            if (f < XPIX * CTI / 4)
                *reinterpret_cast<float4*>(xbuf + buf * XPIX * CPI + (f / (CTI / 4)) * CPI + (f % (CTI / 4)) * 4) = xreg[s];
        }
#pragma unroll
        for (int s = 0; s < YSLOTS; ++s) {
            const int f = tid + s * NT;
            if (f < PTH * PTW * CT / 4)
                *reinterpret_cast<float4*>(ybuf + buf * PTH * PTW * CP + (f / (CT / 4)) * CP + (f % (CT / 4)) * 4) = yreg[s];
        }
    };

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) load_tile(t + 1);
        const float* xb = xbuf + buf * XPIX * CPI + wi * 32 + l31;
        const float* yb = ybuf + buf * PTH * PTW * CP + wo * 32 + l31;
        // k runs over pixel pairs (px, px+1): lanes < 32 carry the even pixel, lanes >= 32 the odd one
#pragma unroll 2
        for (int py = 0; py < PTH; ++py) {
#pragma unroll
            for (int pxp = 0; pxp < PTW / 2; ++pxp) {
                const int px = 2 * pxp + hi;
                const float b = yb[(py * PTW + px) * CP];
                if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, b, accb, 0, 0, 0);
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int r = tap / 3, s = tap - 3 * r;
                    const float av = xb[((py + (TAPS == 9 ? r : 0)) * PW + px + (TAPS == 9 ? s : 0)) * CPI];
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[tap], 0, 0, 0);
                }
            }
        }
        if (t + 1 < t_end) store_tile(buf ^ 1);
        __syncthreads();
    }

    // partial tile -> workspace [split][tap][ci][co]
    float* wsb = a.ws + (size_t)blockIdx.z * TAPS * a.CinP * a.CoutP;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int co = co0 + wo * 32 + l31;
            wsb[((size_t)tap * a.CinP + ci) * a.CoutP + co] = acc[tap][r];
        }
    if (do_bias && hi == 0) a.wsb[(size_t)blockIdx.z * a.CoutP + co0 + wo * 32 + l31] = accb[0];   // row 0 of the tile
}

// 3x3 layers with ONE thin side -- the stem (4 padded input channels) and the head (3 output channels) -- on the vector ALU: on the
// kernel above they fill 4 of 64 ci rows (or 3 of 64 co columns) of every MFMA and took 0.23 / 0.19 ms of a training step for 0.9 /
// 0.7 GFLOP.  MODE 0: thin = Cin (= 4), wide = Cout; MODE 1: thin = Cout (<= 4), wide = Cin.  A workgroup owns 128 wide channels (one
// per lane, two pixel halves) of 8x8-pixel tiles: the wide tensor's tile sits in LDS channel-contiguous (conflict-free), the thin one
// as one float4 per pixel (broadcast reads); 9 taps x 4 thin channels = 36 accumulators per thread.  Partial sums per K split go to the
// same workspace layout [split][tap][CinP][CoutP] (CinP = 4 resp. CoutP = 4) and wgrad_reduce_kernel adds them in a fixed order.
template <int MODE>
__global__ void __launch_bounds__(256) conv_wgrad_thin_f32(const WgradArgs a) {
    constexpr int PW = PTW + 2, PH = PTH + 2;
    constexpr int WPIX = MODE == 0 ? PTH * PTW : PW * PH;         // pixels of the wide tile (dY tile | X halo patch)
    constexpr int TPIX = MODE == 0 ? PW * PH : PTH * PTW;         // pixels of the thin tile
    __shared__ __attribute__((aligned(16))) float wide[WPIX * 128];
    __shared__ float4 thin[TPIX];
    const int tid = threadIdx.x, wc = tid & 127, ph = tid >> 7;
    const int c0 = blockIdx.x * 128;                              // first wide channel of this workgroup
    const int Cw = MODE == 0 ? a.Cout : a.Cin;
    const int t_begin = blockIdx.z * a.tiles_per_split;
    const int t_end = min(a.ptiles, t_begin + a.tiles_per_split);
    float acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
    float accb[4] = {0.f, 0.f, 0.f, 0.f};                         // bias gradient: MODE 0 uses [0] (own co), MODE 1 all (lane 0)

    for (int t = t_begin; t < t_end; ++t) {
        const int n = t / (a.tilesX * a.tilesY);
        const int rem = t - n * (a.tilesX * a.tilesY);
        const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        const int h0 = ty * PTH, w0 = tx * PTW;
        __syncthreads();                                          // the previous tile's readers are done
        for (int f = tid; f < WPIX * 32; f += 256) {
            const int pp = f >> 5, c = (f & 31) * 4;
            int h, w;
            if (MODE == 0) { h = h0 + pp / PTW; w = w0 + pp % PTW; }
            else { h = h0 + pp / PW - 1; w = w0 + pp % PW - 1; }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h >= 0 && h < a.H && w >= 0 && w < a.W && c0 + c < Cw) {
                const size_t pix = (size_t)(n * a.H + h) * a.W + w;
                v = *reinterpret_cast<const float4*>(MODE == 0 ? a.dy + pix * a.ldy + c0 + c : a.x + pix * a.ldx + c0 + c);
            }
            *reinterpret_cast<float4*>(&wide[pp * 128 + c]) = v;
        }
        if (tid < TPIX) {
            int h, w;
            if (MODE == 0) { h = h0 + tid / PW - 1; w = w0 + tid % PW - 1; }
            else { h = h0 + tid / PTW; w = w0 + tid % PTW; }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h >= 0 && h < a.H && w >= 0 && w < a.W) {
                const size_t pix = (size_t)(n * a.H + h) * a.W + w;
                if (MODE == 0) {
                    v = *reinterpret_cast<const float4*>(a.x + pix * a.ldx);
                } else {
                    const float* p = a.dy + pix * a.ldy;
                    v.x = p[0];
                    if (a.Cout > 1) v.y = p[1];
                    if (a.Cout > 2) v.z = p[2];
                    if (a.Cout > 3) v.w = p[3];
                }
            }
            thin[tid] = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int q = 0; q < PTH * PTW / 2; ++q) {
            const int pp = ph * (PTH * PTW / 2) + q, py = pp / PTW, px = pp % PTW;
            if (MODE == 0) {
                const float d = wide[pp * 128 + wc];
                accb[0] += d;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float4 xv = thin[(py + tap / 3) * PW + px + tap % 3];
                    acc[tap][0] = fmaf(xv.x, d, acc[tap][0]);
                    acc[tap][1] = fmaf(xv.y, d, acc[tap][1]);
                    acc[tap][2] = fmaf(xv.z, d, acc[tap][2]);
                    acc[tap][3] = fmaf(xv.w, d, acc[tap][3]);
                }
            } else {
                const float4 d = thin[pp];
                accb[0] += d.x; accb[1] += d.y; accb[2] += d.z; accb[3] += d.w;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float xv = wide[((py + tap / 3) * PW + px + tap % 3) * 128 + wc];
                    acc[tap][0] = fmaf(xv, d.x, acc[tap][0]);
                    acc[tap][1] = fmaf(xv, d.y, acc[tap][1]);
                    acc[tap][2] = fmaf(xv, d.z, acc[tap][2]);
                    acc[tap][3] = fmaf(xv, d.w, acc[tap][3]);
                }
            }
        }
    }
    // the two pixel halves are added through LDS (upper half parks its sums), then the partial tile goes to the workspace
    __syncthreads();
    float* park = wide;                                           // [40][128]
    if (ph == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) park[(t * 4 + c) * 128 + wc] = acc[t][c];
#pragma unroll
        for (int c = 0; c < 4; ++c) park[(36 + c) * 128 + wc] = accb[c];
    }
    __syncthreads();
    if (ph == 1 || c0 + wc >= Cw) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] += park[(t * 4 + c) * 128 + wc];
#pragma unroll
    for (int c = 0; c < 4; ++c) accb[c] += park[(36 + c) * 128 + wc];
    float* wsp = a.ws + (size_t)blockIdx.z * 9 * a.CinP * a.CoutP;
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) wsp[((size_t)t * a.CinP + c) * a.CoutP + c0 + wc] = acc[t][c];
        if (a.wsb) a.wsb[(size_t)blockIdx.z * a.CoutP + c0 + wc] = accb[0];
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            *reinterpret_cast<float4*>(&wsp[((size_t)t * a.CinP + c0 + wc) * a.CoutP]) =
                make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if (a.wsb && c0 + wc == 0)
            *reinterpret_cast<float4*>(&a.wsb[(size_t)blockIdx.z * a.CoutP]) = make_float4(accb[0], accb[1], accb[2], accb[3]);
    }
}

// dW[co][ci][tap] = sum_splits ws[s][tap][ci][co] (+ db[co] = sum_splits wsb[s][co]).  16 outputs x 16 split lanes per workgroup,
// lanes combined in a fixed order: a thread per output walking all splits was one dependent-latency chain of up to 384 loads
// (0.23 ms for the stem's 4608 outputs).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           const float* __restrict__ wsb, float* __restrict__ db, int splits,
                                                           int taps, int Cin, int Cout, int CinP, int CoutP) {
    __shared__ float red[16][17];
    const size_t total = (size_t)Cout * Cin * taps;
    const size_t outputs = total + (db ? Cout : 0);
    const size_t stride = (size_t)taps * CinP * CoutP;
    const int o = threadIdx.x & 15, zl = threadIdx.x >> 4;
    for (size_t base = (size_t)blockIdx.x * 16; base < outputs; base += (size_t)gridDim.x * 16) {
        const size_t i = base + o;
        const float* p = nullptr;
        size_t st = 0, dst = 0;
        bool bias = false;
        if (i < total) {                                   // co fastest: workspace reads are coalesced
            const int co = (int)(i % Cout);
            const size_t t = i / Cout;
            const int ci = (int)(t % Cin);
            const int tap = (int)(t / Cin);
            p = ws + ((size_t)tap * CinP + ci) * CoutP + co;
            st = stride;
            dst = ((size_t)co * Cin + ci) * taps + tap;
        } else if (i < outputs) {
            p = wsb + (i - total);
            st = CoutP;
            dst = i - total;
            bias = true;
        }
        float s = 0.f;
        if (p) {
            int k = zl;
            for (; k + 48 < splits; k += 64)
                s += (p[(size_t)k * st] + p[(size_t)(k + 16) * st]) + (p[(size_t)(k + 32) * st] + p[(size_t)(k + 48) * st]);
            for (; k < splits; k += 16) s += p[(size_t)k * st];
        }
        red[zl][o] = s;
        __syncthreads();
        if (zl == 0 && p) {
            float v[16];
#pragma unroll
            for (int z = 0; z < 16; ++z) v[z] = red[z][o];
            const float r = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                            (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
            if (bias) db[dst] = r;
            else dw[dst] = r;
        }
        __syncthreads();
    }
}

// 1x1 layers on the TN GEMM (winograd_wgrad.hip: gemm_tn_f32): dW[co][ci] = sum_z dU[z][ci][co], transposed through LDS.  The K
// splits are spread over four z lanes per output quad (8 ci x 32 co per workgroup, 16-byte loads, four in flight per thread; lanes
// combined in a fixed order -- deterministic): a 128 -> 256 layer has 256 splits x 128 KB to add, which one 4-byte load chain per
// thread in 32 workgroups took 0.24 ms over.
__global__ void __launch_bounds__(256) tn_finish4_kernel(const float* __restrict__ dU, int splits, float* __restrict__ dw, int Cin,
                                                         int Cout, const float* __restrict__ bias_part, float* __restrict__ db,
                                                         int bias_blocks, int bias_splits) {
    __shared__ float4 red[4][64];
    __shared__ float tile[32][9];
    const int tid = threadIdx.x, o = tid & 63, q = o & 7, cil = o >> 3, zl = tid >> 6;
    if ((int)blockIdx.x < bias_blocks) {             // db[c] = sum_z bias_part[z][c]: 256 columns per workgroup, the same four z lanes
        const int c = (int)blockIdx.x * 256 + o * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < Cout)
            for (int z = zl; z < bias_splits; z += 4) {
                const float4 v = *reinterpret_cast<const float4*>(bias_part + (size_t)z * Cout + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[zl][o] = s;
        __syncthreads();
        if (tid < 64 && c < Cout) {
            const float4 a = red[0][o], b = red[1][o], cc = red[2][o], d = red[3][o];
            *reinterpret_cast<float4*>(db + c) = make_float4((a.x + b.x) + (cc.x + d.x), (a.y + b.y) + (cc.y + d.y),
                                                             (a.z + b.z) + (cc.z + d.z), (a.w + b.w) + (cc.w + d.w));
        }
        return;
    }
    const int bid = (int)blockIdx.x - bias_blocks;
    const int tilesCo = (Cout + 31) / 32;
    const int co0 = (bid % tilesCo) * 32, ci0 = (bid / tilesCo) * 8;
    const size_t per = (size_t)Cin * Cout;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ci0 + cil < Cin && co0 + 4 * q < Cout) {
        const float* p = dU + (size_t)(ci0 + cil) * Cout + co0 + 4 * q;
        int z = zl;
        for (; z + 12 < splits; z += 16) {
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)z * per);
            const float4 b = *reinterpret_cast<const float4*>(p + (size_t)(z + 4) * per);
            const float4 c = *reinterpret_cast<const float4*>(p + (size_t)(z + 8) * per);
            const float4 d = *reinterpret_cast<const float4*>(p + (size_t)(z + 12) * per);
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
            s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; z < splits; z += 4) {
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)z * per);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[zl][o] = s;
    __syncthreads();
    if (tid < 64) {
        const float4 a = red[0][o], b = red[1][o], c = red[2][o], d = red[3][o];
        tile[4 * q + 0][cil] = (a.x + b.x) + (c.x + d.x);
        tile[4 * q + 1][cil] = (a.y + b.y) + (c.y + d.y);
        tile[4 * q + 2][cil] = (a.z + b.z) + (c.z + d.z);
        tile[4 * q + 3][cil] = (a.w + b.w) + (c.w + d.w);
    }
    __syncthreads();
    const int c = tid >> 3, k = tid & 7;
    if (co0 + c < Cout && ci0 + k < Cin) dw[(size_t)(co0 + c) * Cin + ci0 + k] = tile[c][k];
}

// ---- 1x1 weight gradients on the bf16x3 pipe kernel (bbdm_gemm_bf3p_tn_f32): its operands are TRANSPOSED bf16 planes, rows = channels,
// k = pixels: [C / 32][KPad / 16][3][1 KB], element (r = c & 31, k = pixel & 15) at (k >> 3) * 512 + r * 16 + (k & 7) * 2.  This pass writes
// them from a plain [pixels][C] fp32 tensor: a workgroup stages 16 pixels x 256 channels in LDS (coalesced rows), then thread c reads its
// channel's 8 + 8 pixels (conflict-free), splits them exactly and stores 3 x 16 B per half -- 32 lanes = 512 contiguous bytes of a
// plane.  Pixels >= K and channels >= C are written as zeros (they enter the contraction / pad the 128-column tiles).  `colsum`: the
// workgroup also adds up its pixels per channel (fp64 per thread) and writes one partial row per pixel walker: the bias gradient of
// the layer when the tensor is dY, added over the walkers by tn_finish4_kernel.
__global__ void __launch_bounds__(256) tn_pack_planes_kernel(const float* __restrict__ src, int ld, long long K, int C, int CPad,
                                                             long long KPad, unsigned char* __restrict__ dst,
                                                             float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) float tile[16][256 + 4];
    const int tid = threadIdx.x, c0 = blockIdx.x * 256;
    const long long kblocks = KPad / 16;
    double acc = 0.0;
    for (long long kb = blockIdx.y; kb < kblocks; kb += gridDim.y) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int f = tid + s * 256, row = f >> 6, c = (f & 63) * 4;
            const long long p = kb * 16 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < K && c0 + c < C) v = *reinterpret_cast<const float4*>(src + (size_t)p * ld + c0 + c);      // (C % 4 == 0)
            *reinterpret_cast<float4*>(&tile[row][c]) = v;
        }
        __syncthreads();
        if (c0 + tid < CPad) {
            const int cb = (c0 + tid) >> 5, r = (c0 + tid) & 31;
            unsigned char* d = dst + (((size_t)cb * kblocks + kb) * 3) * 1024 + r * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tile[8 * h + j][tid];
                if (colsum) acc += (double)(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
                uint2 a1, a2, a3, b1, b2, b3;
                split4(make_float4(v[0], v[1], v[2], v[3]), a1, a2, a3);
                split4(make_float4(v[4], v[5], v[6], v[7]), b1, b2, b3);
                *reinterpret_cast<uint4*>(d + h * 512) = make_uint4(a1.x, a1.y, b1.x, b1.y);
                *reinterpret_cast<uint4*>(d + 1024 + h * 512) = make_uint4(a2.x, a2.y, b2.x, b2.y);
                *reinterpret_cast<uint4*>(d + 2048 + h * 512) = make_uint4(a3.x, a3.y, b3.x, b3.y);
            }
        }
        __syncthreads();
    }
    if (colsum && c0 + tid < C) colsum[(size_t)blockIdx.y * C + c0 + tid] = (float)acc;
}

// db[c] = sum over rows of dy[M][ld]; fp64 per-thread accumulation, then exact integer-limb cells (stats_acc.h: any order of the
// atomics leaves the same limbs -- the bias gradients are bitwise reproducible)
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ dy, int ld, unsigned long long* __restrict__ acc,
                                                     long long M, int C, int rows_per_block) {
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    // thread -> column c = tid % Cb, row lane = tid / Cb
    const int Cb = C < 256 ? C : 256;
    const int RL = 256 / Cb;
    for (int cbase = 0; cbase < C; cbase += 256) {
        const int c = cbase + tid % Cb;
        const int rl = tid / Cb;
        if (c < C && rl < RL) {
            double s = 0.0;
            for (long long r = r0 + rl; r < r1; r += RL) s += (double)dy[(size_t)r * ld + c];
            sa_add(acc + (size_t)c * SA_W, s);
        }
    }
}

// vectorised variant (C % 4 == 0, 16-byte aligned rows): a workgroup owns 256 columns (64 lanes x float4) of a row range, its
// four waves take every fourth row with four loads in flight each; fp64 per-lane accumulators, one LDS reduction over the
// waves, then the limb cells.  (The scalar kernel above keeps ONE 4-byte load in flight per thread: 0.23 ms for a 268 MB dY.)
__global__ void __launch_bounds__(256) colsum4_kernel(const float* __restrict__ dy, int ld, unsigned long long* __restrict__ acc, long long M,
                                                      int C, int rows_per_block) {
    __shared__ double red[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + lane * 4;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < C) {
        const float* p = dy + c;
        long long r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)r * ld);
            const float4 b = *reinterpret_cast<const float4*>(p + (size_t)(r + 4) * ld);
            const float4 d = *reinterpret_cast<const float4*>(p + (size_t)(r + 8) * ld);
            const float4 e = *reinterpret_cast<const float4*>(p + (size_t)(r + 12) * ld);
            s0 += ((double)a.x + (double)b.x) + ((double)d.x + (double)e.x);
            s1 += ((double)a.y + (double)b.y) + ((double)d.y + (double)e.y);
            s2 += ((double)a.z + (double)b.z) + ((double)d.z + (double)e.z);
            s3 += ((double)a.w + (double)b.w) + ((double)d.w + (double)e.w);
        }
        for (; r < r1; r += 4) {
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)r * ld);
            s0 += (double)a.x; s1 += (double)a.y; s2 += (double)a.z; s3 += (double)a.w;
        }
    }
    red[wave][lane][0] = s0; red[wave][lane][1] = s1; red[wave][lane][2] = s2; red[wave][lane][3] = s3;
    __syncthreads();
    if (wave == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            sa_add(acc + (size_t)(c + j) * SA_W, (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]));
    }
}

// per-image variant: grid.y = image
__global__ void __launch_bounds__(256) colsum_batched_kernel(const float* __restrict__ dy, int ld, unsigned long long* __restrict__ acc,
                                                             long long M, int C, int rows_per_block) {
    const int tid = threadIdx.x, n = blockIdx.y;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    const float* base = dy + (size_t)n * M * ld;
    const int Cb = C < 256 ? C : 256;
    const int RL = 256 / Cb;
    for (int cbase = 0; cbase < C; cbase += 256) {
        const int c = cbase + tid % Cb, rl = tid / Cb;
        if (c < C && rl < RL) {
            double s = 0.0;
            for (long long r = r0 + rl; r < r1; r += RL) s += (double)base[(size_t)r * ld + c];
            sa_add(acc + ((size_t)n * C + c) * SA_W, s);
        }
    }
}

__global__ void colsum_batched_final_kernel(const unsigned long long* __restrict__ acc, float* __restrict__ out, int ldo, int N, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * C) out[(size_t)(i / C) * ldo + i % C] = (float)sa_load(acc + (size_t)i * SA_W);
}

__global__ void colsum_final_kernel(const unsigned long long* __restrict__ acc, float* __restrict__ db, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) db[c] = (float)sa_load(acc + (size_t)c * SA_W);
}

// packed weights for the data-gradient conv: Wd[ci_out][co_in][r][s] = W[co_in][ci_out][K-1-r][K-1-s]
__global__ void pack_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int CoutIn,
                                         int ks, int CinPad128, int nchunks) {
    const int KC = 16;
    const size_t total = (size_t)ks * ks * nchunks * CinPad128 * KC;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = i % KC;
        size_t t = i / KC;
        const int ci = t % CinPad128;          // output channel of the dgrad conv
        t /= CinPad128;
        const int chunk = t % nchunks;
        const int tap = t / nchunks;
        const int co = chunk * KC + k;         // input channel of the dgrad conv
        float v = 0.f;
        if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * ks * ks + (ks * ks - 1 - tap)];
        p[i] = v;
    }
}

}  // namespace

// geometry shared by the workspace query and the launcher: ci tile 128 (8 waves) when Cin >= 128, else 64 (4 waves); thin 3x3
// layers (conv_wgrad_thin_f32): mode 0 = 4 input channels, mode 1 = at most 4 output channels
struct WgradGeom {
    int cti, CinP, CoutP, ptiles, splits, tiles_per_split, thin;
};
static int wgrad_thin_mode(int Cin, int Cout, int ks) {
    if (ks != 3) return -1;
    if (Cout <= 4 && Cin % 4 == 0) return 1;
    if (Cin == 4 && Cout % 4 == 0) return 0;
    return -1;
}
static WgradGeom wgrad_geom(int N, int H, int W, int Cin, int Cout, int thin) {
    WgradGeom g;
    g.thin = thin;
    g.cti = Cin >= 128 ? 128 : 64;
    g.CinP = cdiv(Cin, g.cti) * g.cti;
    g.CoutP = cdiv(Cout, CT) * CT;
    g.ptiles = N * cdiv(H, PTH) * cdiv(W, PTW);
    int tiles = (g.CinP / g.cti) * (g.CoutP / CT);
    int want = g.cti == 128 ? 512 : 768;
    if (thin == 0) { g.CinP = 4; tiles = cdiv(Cout, 128); want = 768; }
    if (thin == 1) { g.CoutP = 4; tiles = cdiv(Cin, 128); want = 768; }
    int splits = cdiv(want, tiles);
    if (splits > g.ptiles) splits = g.ptiles;
    if (splits < 1) splits = 1;
    g.tiles_per_split = cdiv(g.ptiles, splits);
    g.splits = cdiv(g.ptiles, g.tiles_per_split);
    return g;
}
static size_t wgrad_geom_floats(const WgradGeom& g, int ks) {
    return (size_t)g.splits * ks * ks * g.CinP * g.CoutP + (size_t)g.splits * g.CoutP;
}

// 1x1 layers wide enough for 128-wide MFMA tiles take the TN GEMM of winograd_wgrad.hip (K = pixels): 100-117 TFLOP/s where
// conv_wgrad_f32<1, 4> (one 32x32 accumulator per wave, 21 FLOP per LDS byte) reaches ~70
static bool wgrad_tn_path(int Cin, int Cout, int ks) { return ks == 1 && Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 64 && Cout >= 64; }

// ... and, where the plane layout takes the shape, on the bf16x3 pipe kernel behind a transposing split pass per operand (10 B per
// element moved once, against a GEMM at 200+ instead of 60-100 TFLOP/s): option wgrad1x1_bf3 = 0 keeps gemm_tn_f32 (A/B runs)
struct TnPlanes {
    long long KPad;
    size_t at_bytes, bt_bytes;
    int splits, walkers;
};
static bool wgrad_tn_planes(long long K, int Cin, int Cout, TnPlanes* g) {
    const int on = bbdm_option(BBDM_OPT_WGRAD1X1_BF3);
    if (!on || Cin % 32 != 0 || Cout % 4 != 0 || K < 4096) return false;
    // the split passes move 10 B per element of X and dY once: they pay where the GEMM has >= ~100 FLOP per such byte (measured on the
    // LBBDM-f4 step, batch 32: 2048 -> 1024 0.42 -> 0.28 ms, 1024 -> 3072 0.48 -> 0.40; 128 -> 256 at 131072 pixels 0.12 -> 0.26);
    // wgrad1x1_bf3 = 2 takes every shape the layout accepts (tests)
    if (on < 2 && (long long)Cin * Cout < 512ll * (Cin + Cout)) return false;
    const long long KPad = (K + 255) / 256 * 256;
    if (!bbdm_gemm_bf3p_tn_supported(KPad, Cin, Cout)) return false;
    if (g) {
        g->KPad = KPad;
        g->at_bytes = bbdm_gemm_bf3p_tn_at_bytes(1, KPad, Cin);
        g->bt_bytes = bbdm_gemm_bf3p_tn_bt_bytes(1, KPad, Cout);
        g->splits = bbdm_gemm_bf3p_tn_splits(1, KPad, Cin, Cout);
        const long long kb = KPad / 16;
        g->walkers = (int)(kb < 512 ? kb : 512);
    }
    return true;
}
static size_t wgrad_tn_planes_floats(const TnPlanes& g, int Cin, int Cout) {
    return (g.at_bytes + g.bt_bytes) / 4 + (size_t)g.splits * Cin * Cout + 4 + (size_t)g.walkers * Cout + 4;
}

static size_t wgrad_tn_floats(long long K, int Cin, int Cout) {
    const size_t sp = (size_t)bbdm_gemm_tn_splits(1, K, Cin, Cout);
    return sp * Cin * Cout + 4 + (sp > 2 * SA_W ? sp : 2 * SA_W) * (size_t)Cout + 2;      // partial tiles | per-split column sums / colsum limb cells
}

extern "C" size_t bbdm_conv_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int ks) {
    size_t need = wgrad_geom_floats(wgrad_geom(N, H, W, Cin, Cout, -1), ks);      // the thin path may be refused at launch (pitches)
    const int thin = wgrad_thin_mode(Cin, Cout, ks);
    if (thin >= 0) {
        const size_t t = wgrad_geom_floats(wgrad_geom(N, H, W, Cin, Cout, thin), ks);
        if (t > need) need = t;
    }
    TnPlanes tp;
    if (wgrad_tn_path(Cin, Cout, ks) && wgrad_tn_planes((long long)N * H * W, Cin, Cout, &tp)) {
        const size_t t = wgrad_tn_planes_floats(tp, Cin, Cout);
        if (t > need) need = t;
    }
    if (wgrad_tn_path(Cin, Cout, ks)) {
        const size_t tn = wgrad_tn_floats((long long)N * H * W, Cin, Cout);
        if (tn > need) need = tn;
    }
    return need;
}

template <int TAPS, int WI>
static int launch_wgrad(const WgradArgs& a, dim3 grid, hipStream_t st) {
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int XPIX = (PTW + 2 * PAD) * (PTH + 2 * PAD);
    constexpr size_t lds = ((size_t)2 * XPIX * (WI * 32 + 4) + 2 * PTH * PTW * CP) * sizeof(float);
    static bool attr_set_dev[BBDM_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[bbdm_device_slot()];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_f32<TAPS, WI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            bbdm_set_error("conv_wgrad: hipFuncSetAttribute(%zu B LDS) failed", lds);
            return BBDM_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_wgrad_f32<TAPS, WI>), grid, dim3(WI * 128), lds, st, a);
    return 0;
}

extern "C" int bbdm_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias,
                                   float* ws, size_t ws_floats, int N, int H, int W, int Cin, int Cout, int ks, void* stream) {
    BBDM_REQUIRE(x && dy && dw_oihw && ws, "conv_wgrad: null pointer");
    BBDM_REQUIRE(ks == 1 || ks == 3, "conv_wgrad: ks=%d", ks);
    BBDM_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv_wgrad: bad shape");
    BBDM_REQUIRE(ldx % 4 == 0 && ldx >= Cin && ((uintptr_t)x & 15) == 0, "conv_wgrad: x pitch/alignment (ldx=%d)", ldx);
    BBDM_REQUIRE(ldy >= Cout && (Cout % 4 != 0 || (ldy % 4 == 0 && ((uintptr_t)dy & 15) == 0)),
                 "conv_wgrad: dy pitch/alignment (ldy=%d)", ldy);
    BBDM_REQUIRE(Cin % 4 == 0, "conv_wgrad: Cin=%d must be a multiple of 4 (pad the input tensor)", Cin);
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_tn_path(Cin, Cout, ks) && ((uintptr_t)dy & 15) == 0 && ldy % 4 == 0 && ((uintptr_t)ws & 15) == 0) {
        const long long K = (long long)N * H * W;
        TnPlanes tp;
        // (a workspace too small for the plane path -- sized under another wgrad1x1_bf3 setting, say -- takes the TN GEMM below)
        if (wgrad_tn_planes(K, Cin, Cout, &tp) && ws_floats >= wgrad_tn_planes_floats(tp, Cin, Cout) && ((uintptr_t)dw_oihw & 3) == 0 &&
            (!dbias || ((uintptr_t)dbias & 15) == 0)) {
            unsigned char* at = reinterpret_cast<unsigned char*>(ws);
            unsigned char* bt = at + tp.at_bytes;
            float* dU = reinterpret_cast<float*>(bt + tp.bt_bytes);
            float* bias_part = dbias ? dU + (((size_t)tp.splits * Cin * Cout + 3) & ~(size_t)3) : nullptr;
            const int CoutPad = cdiv(Cout, 128) * 128;
            hipLaunchKernelGGL(tn_pack_planes_kernel, dim3(cdiv(Cin, 256), tp.walkers), dim3(256), 0, st, x, ldx, K, Cin, Cin, tp.KPad,
                               at, (float*)nullptr);
            hipLaunchKernelGGL(tn_pack_planes_kernel, dim3(cdiv(CoutPad, 256), tp.walkers), dim3(256), 0, st, dy, ldy, K, Cout,
                               CoutPad, tp.KPad, bt, bias_part);
            BBDM_CHECK_LAUNCH("conv_wgrad(tn planes: pack)");
            int rc = bbdm_gemm_bf3p_tn_f32(at, bt, dU, 1, tp.KPad, Cin, Cout, stream);
            if (rc != BBDM_OK) return rc;
            const int bias_blocks = bias_part ? cdiv(Cout, 256) : 0;
            hipLaunchKernelGGL(tn_finish4_kernel, dim3((unsigned)(cdiv(Cout, 32) * cdiv(Cin, 8) + bias_blocks)), dim3(256), 0, st, dU,
                               tp.splits, dw_oihw, Cin, Cout, bias_part, dbias, bias_blocks, tp.walkers);
            BBDM_CHECK_LAUNCH("conv_wgrad(tn planes)");
            return BBDM_OK;
        }
        BBDM_REQUIRE(ws_floats >= wgrad_tn_floats(K, Cin, Cout), "conv_wgrad: workspace of %zu floats, the TN-GEMM path needs %zu",
                     ws_floats, wgrad_tn_floats(K, Cin, Cout));
        const int splits = bbdm_gemm_tn_splits(1, K, Cin, Cout);
        // the bias gradient rides along: per-split column sums of dY from the GEMM's B staging, added over the splits by the finish
        // kernel (bbdm_colsum_f32 was three more launches and a second pass over dY per layer)
        float* bias_part = nullptr;
        if (dbias && ((uintptr_t)dbias & 15) == 0) bias_part = ws + (((size_t)splits * Cin * Cout + 3) & ~(size_t)3);
        int rc = bbdm_gemm_tn_impl(x, ldx, 0, dy, ldy, 0, ws, bias_part, 1, K, Cin, Cout, stream);
        if (rc != BBDM_OK) return rc;
        const int bias_blocks = bias_part ? cdiv(Cout, 256) : 0;
        hipLaunchKernelGGL(tn_finish4_kernel, dim3((unsigned)(cdiv(Cout, 32) * cdiv(Cin, 8) + bias_blocks)), dim3(256), 0, st, ws,
                           splits, dw_oihw, Cin, Cout, bias_part, dbias, bias_blocks, splits);
        BBDM_CHECK_LAUNCH("conv_wgrad(tn)");
        if (dbias && !bias_part) {
            size_t off = ((size_t)splits * Cin * Cout + 1) & ~(size_t)1;        // 8-byte alignment of the fp64 scratch
            rc = bbdm_colsum_f32(dy, ldy, reinterpret_cast<double*>(ws + off), dbias, K, Cout, stream);
        }
        return rc;
    }
    int thin = wgrad_thin_mode(Cin, Cout, ks);
    if (thin == 0 && (ldy % 4 != 0 || ((uintptr_t)dy & 15) != 0)) thin = -1;      // float4 loads of the wide tensor
    if (thin >= 0 && ((uintptr_t)ws & 15) != 0) thin = -1;
    const WgradGeom g = wgrad_geom(N, H, W, Cin, Cout, thin);
    BBDM_REQUIRE(ws_floats >= wgrad_geom_floats(g, ks), "conv_wgrad: workspace of %zu floats, this shape needs %zu", ws_floats,
                 wgrad_geom_floats(g, ks));
    WgradArgs a;
    a.x = x; a.dy = dy; a.ws = ws; a.ldx = ldx; a.ldy = ldy;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.taps = ks * ks; a.pad = ks / 2;
    a.tilesX = cdiv(W, PTW); a.tilesY = cdiv(H, PTH);
    a.ptiles = g.ptiles;
    a.CinP = g.CinP; a.CoutP = g.CoutP;
    a.tiles_per_split = g.tiles_per_split;
    const int splits = g.splits;
    a.wsb = dbias ? ws + (size_t)splits * ks * ks * a.CinP * a.CoutP : nullptr;
    int rc = 0;
    if (thin == 0) {
        hipLaunchKernelGGL((conv_wgrad_thin_f32<0>), dim3(cdiv(Cout, 128), 1, splits), dim3(256), 0, st, a);
    } else if (thin == 1) {
        hipLaunchKernelGGL((conv_wgrad_thin_f32<1>), dim3(cdiv(Cin, 128), 1, splits), dim3(256), 0, st, a);
    } else {
        const dim3 grid(a.CinP / g.cti, a.CoutP / CT, splits);
        if (ks == 3) rc = g.cti == 128 ? launch_wgrad<9, 4>(a, grid, st) : launch_wgrad<9, 2>(a, grid, st);
        else rc = g.cti == 128 ? launch_wgrad<1, 4>(a, grid, st) : launch_wgrad<1, 2>(a, grid, st);
    }
    if (rc != 0) return rc;
    const size_t total = (size_t)Cout * Cin * ks * ks;
    int rb = (int)((total + Cout + 15) / 16);
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, ws, dw_oihw, a.wsb, dbias, splits, ks * ks, Cin,
                       Cout, a.CinP, a.CoutP);
    BBDM_CHECK_LAUNCH("conv_wgrad");
    return BBDM_OK;
}

extern "C" int bbdm_colsum_f32(const float* dy, int ld, double* acc_, float* out, long long M, int C, void* stream) {
    BBDM_REQUIRE(dy && acc_ && out && M > 0 && C > 0 && ld >= C && ((uintptr_t)acc_ & 7) == 0, "colsum: bad args");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(acc_);       // [C][SA_W] limb cells
    bbdm_zero_async(acc, 8 * (size_t)C * SA_W, st);
    long long blocks = (M + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const int rpb = (int)((M + blocks - 1) / blocks);
    blocks = (M + rpb - 1) / rpb;
    if (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)dy & 15) == 0) {
        const int cchunks = cdiv(C, 256);
        long long rb = 2048 / cchunks;                                // ~8 workgroups per CU in total
        if (rb > (M + 63) / 64) rb = (M + 63) / 64;                   // >= 64 rows (16 per wave) per workgroup
        if (rb < 1) rb = 1;
        const int rows = (int)((M + rb - 1) / rb);
        rb = (M + rows - 1) / rows;
        hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)rb, cchunks), dim3(256), 0, st, dy, ld, acc, M, C, rows);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, ld, acc, M, C, rpb);
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, acc, out, C);
    BBDM_CHECK_LAUNCH("colsum");
    return BBDM_OK;
}

extern "C" int bbdm_colsum_batched_f32(const float* dy, int ld, double* acc_, float* out, int ldo, int N, long long M, int C,
                                       void* stream) {
    BBDM_REQUIRE(dy && acc_ && out && N > 0 && M > 0 && C > 0 && ld >= C && ldo >= C && ((uintptr_t)acc_ & 7) == 0, "colsum_batched: bad args");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(acc_);       // [N][C][SA_W] limb cells
    bbdm_zero_async(acc, 8 * (size_t)N * C * SA_W, st);
    long long blocks = (M + 255) / 256;
    const long long cap = cdiv(1024, N) > 1 ? cdiv(1024, N) : 1;
    if (blocks > cap) blocks = cap;
    const int rpb = (int)((M + blocks - 1) / blocks);
    blocks = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_batched_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, dy, ld, acc, M, C, rpb);
    hipLaunchKernelGGL(colsum_batched_final_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, st, acc, out, ldo, N, C);
    BBDM_CHECK_LAUNCH("colsum_batched");
    return BBDM_OK;
}

extern "C" size_t bbdm_conv_packed_dgrad_floats(int Cout, int Cin, int CoutIn, int ks) {
    (void)Cout;
    const int CinPad128 = cdiv(Cin, 128) * 128;
    const int nchunks = cdiv(CoutIn, 16);
    return (size_t)ks * ks * nchunks * CinPad128 * 16;
}

extern "C" int bbdm_conv_pack_weight_dgrad_f32(const float* w_oihw, float* packed, int Cout, int Cin, int CoutIn, int ks,
                                               void* stream) {
    BBDM_REQUIRE(w_oihw && packed, "conv_pack_dgrad: null pointer");
    BBDM_REQUIRE((ks == 1 || ks == 3) && Cout > 0 && Cin > 0 && CoutIn >= Cout && CoutIn % 4 == 0,
                 "conv_pack_dgrad: bad shape Cout=%d Cin=%d CoutIn=%d ks=%d", Cout, Cin, CoutIn, ks);
    const int CinPad128 = cdiv(Cin, 128) * 128;
    const int nchunks = cdiv(CoutIn, 16);
    const size_t total = (size_t)ks * ks * nchunks * CinPad128 * 16;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout, Cin,
                       CoutIn, ks, CinPad128, nchunks);
    BBDM_CHECK_LAUNCH("conv_pack_dgrad");
    return BBDM_OK;
}
