// winograd_wgrad.hip -- weight gradient of the 3x3 stride-1 convolutions in the Winograd domain (training path).
//
// Reference: autograd of nn.Conv2d at openaimodel.py:207,233,524,690 (ATen's MIOpen/cuDNN backward-weights); same
// layers as winograd.hip serves in the forward / data-gradient direction.
//
//   forward   Y  = A^T [ U (.) V ] A,   U = G g G^T,   V = B^T d B        (per tile; summed over input channels)
//   =>        dU_xi[ci][co] = sum_tiles V_xi[t][ci] * dM_xi[t][co],   dM = A dY A^T          ((m+2)^2 GEMMs, K = tiles)
//             dg = G^T dU G
//
// (m+2)^2 multiplies per m^2 pixels instead of 9 m^2: 2.25x (m = 2), 4x (m = 4) or 5.06x (m = 6) fewer FLOPs than the
// direct weight gradient (conv_wgrad.hip), which was 49 % of the C4 training step's kernel time.  Four launches:
//   (1) winograd_input_kernel (winograd.hip, no fused producer): x NHWC -> V[(m+2)^2][tiles][Cin]          HBM-bound
//   (2) winograd_dy_kernel   : dY NHWC -> dM[(m+2)^2][tiles][Cout]                                          HBM-bound
//   (3) gemm_tn_f32          : dU[split][xi] = V_xi^T dM_xi over a K range of tiles, fp32 MFMA              MFMA-bound
//   (4) wgrad_finish_kernel  : sum the K splits in a fixed order (deterministic), G^T . G, write OIHW
// The bias gradient = column sums of the ONE plane dM_(1,1) (row 1 of A is all ones: that plane holds the tile sums of dY).
// fp32 throughout; rounding error of the gradient relative to fp64 (tests/test_winograd_math_cpu.py): direct 5e-7,
// m = 2: 6e-7, m = 4: 3e-6, m = 6: 5e-6.
// m = 8 (end of round 5; ten points, winograd_math.h): only on the bf16x3 pipeline -- the transposed planes of V from the training forward's
// input transform, winograd_dy_split_kernel<8> (csrc/winograd.hip), bbdm_gemm_bf3p_tn_f32, and the finish kernels below with G^T . G in fp64
// (3e-5); stages (1) - (3) of the fp32 form keep m <= 6.
#include "winograd_math.h"

namespace {

// ---- (2) dY transform: one thread = one (tile, channel vector) ---------------------------------------------------------
// VT = float4 (m = 2, 4) or float2 (m = 6: 8 x 6 intermediate + 8 outputs per row would not fit 4-wide).
template <int MO, typename VT>
__global__ void __launch_bounds__(256) winograd_dy_kernel(const float* __restrict__ dy, int ld, float* __restrict__ dM, int N,
                                                          int H, int W, int C, size_t plane) {
    constexpr int AL = MO + 2;
    constexpr int VW = sizeof(VT) / sizeof(float);
    const int CV = C / VW;
    const int TH = (H + MO - 1) / MO, TW = (W + MO - 1) / MO;
    const long long total = (long long)N * TH * TW * CV;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256) {
        const int c = (int)(u % CV) * VW;
        const long long tile = u / CV;
        const int tw = (int)(tile % TW);
        const long long r = tile / TW;
        const int th = (int)(r % TH), n = (int)(r / TH);
        VT t[AL][MO];                       // t[i][j] = (A dY)[i][j]
#pragma unroll
        for (int j = 0; j < MO; ++j) {
            // one column at a time, its loads back to back; pixels beyond the image (ragged m = 6 tiles) read a clamped address
            // and are zeroed by a 0/1 factor (a bounds-check branch per load would serialise them: see winograd_input6_kernel)
            const int w = MO * tw + j;
            const int wc = min(w, W - 1);
            const float wmask = w < W ? 1.f : 0.f;
            VT d[MO], col[AL];
#pragma unroll
            for (int i = 0; i < MO; ++i) {
                const int hc = min(MO * th + i, H - 1);
                d[i] = *reinterpret_cast<const VT*>(dy + ((size_t)(n * H + hc) * W + wc) * ld + c);
            }
#pragma unroll
            for (int i = 0; i < MO; ++i) d[i] = ((MO * th + i < H) ? wmask : 0.f) * d[i];
            a_transform<MO>(d, col);
#pragma unroll
            for (int i = 0; i < AL; ++i) t[i][j] = col[i];
            __builtin_amdgcn_sched_barrier(0);
        }
        float* o = dM + (size_t)tile * C + c;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            VT row[AL];
            a_transform<MO>(t[i], row);
#pragma unroll
            for (int j = 0; j < AL; ++j) {      // a walking pointer + one row at a time: computing the (m+2)^2 plane addresses
                *reinterpret_cast<VT*>(o) = row[j];     // up front costs 2 VGPRs each (346 instead of 170 for m = 6)
                o += plane;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- (3) batched TN GEMM: C[z][b][m][n] = sum_{k in split z} A[b][k][m] * B[b][k][n] ------------------------------------
// Both operands have the contraction index as their SLOW memory dimension (rows = tiles), which is what
// v_mfma_f32_32x32x2_f32 wants: lane l supplies A[k = l >> 5][m = l & 31] and B[k = l >> 5][n = l & 31], i.e. 32 consecutive
// floats of one LDS row per half-wave -- the tiles are staged as plain row copies, no transpose anywhere.
// Workgroup = WM x 2 waves, tile (64 WM) x 128, each wave 64 x 64 = 2 x 2 MFMA tiles; K walked 16 rows per stage, register
// prefetch of the next stage + double-buffered LDS (one barrier per stage).  LDS pitches are = 32 (mod 64) floats so that the
// two half-waves (rows k and k + 1) of a fragment read hit disjoint banks.
constexpr int TKC = 16;
constexpr int TBN = 128;

struct TnArgs {
    const float* A;
    const float* B;
    float* C;
    size_t a_stride, b_stride;
    int lda, ldb;
    int M, N;
    long long K;
    int k_per_split;             // multiple of TKC
    int batch, splits;
    int tilesM, tilesN;
    int items;                   // batch * splits
    float* bias_part;            // [splits][batch][N] column sums of B over each split's K range (tiles mt == 0 write them), or null
};

template <int WM>
__global__ void __launch_bounds__(WM * 128, 2) gemm_tn_f32(const TnArgs a) {
    constexpr int BM = WM * 64, NT = WM * 128;
    constexpr int PA = BM + 32, PB = TBN + 32;
    constexpr int ASLOTS = TKC * BM / 4 / NT;            // 2
    constexpr int BSLOTS = TKC * TBN / 4 / NT;           // 2 (WM = 2) or 1 (WM = 4)
    constexpr int STAGE = TKC * (PA + PB);
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][TKC][PA] A rows, then [TKC][PB] B rows per stage

    // Workgroup ids are dealt round-robin to the 8 XCDs: all tiles of one (batch, split) item go to ONE XCD, so that the
    // A / B panels every tile of the item re-reads are shared through that XCD's L2.
    const int L = (int)blockIdx.x, xcd = L & 7, j = L >> 3;
    const int tpi = a.tilesM * a.tilesN;
    const int item = (j / tpi) * 8 + xcd;
    if (item >= a.items) return;
    const int tile = j % tpi;
    const int nt = tile % a.tilesN, mt = tile / a.tilesN;
    const int b = item % a.batch, z = item / a.batch;
    const int m0 = mt * BM, n0 = nt * TBN;
    const long long k_begin = (long long)z * a.k_per_split;
    const long long k_end = k_begin + a.k_per_split < a.K ? k_begin + a.k_per_split : a.K;
    const float* __restrict__ Ab = a.A + (size_t)b * a.a_stride;
    const float* __restrict__ Bb = a.B + (size_t)b * a.b_stride;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    float4 areg[1][ASLOTS], breg[1][BSLOTS];
    // column sums of B (the bias gradient of a 1x1 layer: B = dY) ride along: a thread's B slots all hold the same column quad, so it
    // adds them up as they pass through its registers -- no second pass over dY
    const bool do_bias = a.bias_part != nullptr && mt == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](long long k0, float4 (&ar)[ASLOTS], float4 (&br)[BSLOTS]) {
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s) {
            const int f = tid + s * NT;
            const int k = f / (BM / 4), c = (f % (BM / 4)) * 4;
            // clamped (always valid) address + select: a bounds-check branch per load would make the waitcnt pass drain vmcnt
            // between the loads of a stage
            const long long kr = k0 + k < a.K ? k0 + k : a.K - 1;
            const int mc = m0 + c < a.M ? m0 + c : a.M - 4;
            const float4 v = *reinterpret_cast<const float4*>(Ab + (size_t)kr * a.lda + mc);
            ar[s] = (k0 + k < k_end && m0 + c < a.M) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < BSLOTS; ++s) {
            const int f = tid + s * NT;
            const int k = f / (TBN / 4), c = (f % (TBN / 4)) * 4;
            const long long kr = k0 + k < a.K ? k0 + k : a.K - 1;
            const int nc = n0 + c < a.N ? n0 + c : a.N - 4;
            const float4 v = *reinterpret_cast<const float4*>(Bb + (size_t)kr * a.ldb + nc);
            br[s] = (k0 + k < k_end && n0 + c < a.N) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store = [&](float* st, const float4 (&ar)[ASLOTS], const float4 (&br)[BSLOTS]) {
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s) {
            const int f = tid + s * NT;
            *reinterpret_cast<float4*>(st + (f / (BM / 4)) * PA + (f % (BM / 4)) * 4) = ar[s];
        }
#pragma unroll
        for (int s = 0; s < BSLOTS; ++s) {
            const int f = tid + s * NT;
            *reinterpret_cast<float4*>(st + TKC * PA + (f / (TBN / 4)) * PB + (f % (TBN / 4)) * 4) = br[s];
            if (do_bias) { bsum.x += br[s].x; bsum.y += br[s].y; bsum.z += br[s].z; bsum.w += br[s].w; }
        }
    };
    auto multiply = [&](int chunk) {
        const float* sA = smem + (chunk & 1) * STAGE + wm * 64 + l31;
        const float* sB = smem + (chunk & 1) * STAGE + TKC * PA + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < TKC / 2; ++kk) {
            const int k = 2 * kk + hi;
            const float a0 = sA[k * PA], a1 = sA[k * PA + 32];
            const float b0 = sB[k * PB], b1 = sB[k * PB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };

    // register prefetch of the next chunk + double-buffered LDS, one barrier per chunk.  (A second register set -- chunk c + 2 requested
    // while c is multiplied -- was measured in round 3: 92 instead of 56 VGPRs and 15 % SLOWER on the training step's 1x1 layers.)
    const int nchunks = (int)((k_end - k_begin + TKC - 1) / TKC);
    if (nchunks > 0) {
        load(k_begin, areg[0], breg[0]);
        store(smem, areg[0], breg[0]);
    }
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) load(k_begin + (long long)(chunk + 1) * TKC, areg[0], breg[0]);
        multiply(chunk);
        if (more) store(smem + ((chunk + 1) & 1) * STAGE, areg[0], breg[0]);
        __syncthreads();
    }

    if (do_bias) {                                   // (workgroup-uniform) the NT / 32 threads of a column quad, in a fixed order
        float4* sb = reinterpret_cast<float4*>(smem);
        sb[tid] = bsum;
        __syncthreads();
        if (tid < 32 && n0 + tid * 4 < a.N) {
            float4 t = sb[tid];
            for (int r = 1; r < NT / 32; ++r) {
                const float4 v = sb[r * 32 + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            *reinterpret_cast<float4*>(a.bias_part + ((size_t)z * a.batch + b) * a.N + n0 + tid * 4) = t;
        }
    }
    float* __restrict__ Cb = a.C + ((size_t)z * a.batch + b) * (size_t)a.M * a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int n = n0 + wn * 64 + jj * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < a.M && n < a.N) Cb[(size_t)m * a.N + n] = acc[i][jj][r];
            }
        }
}

// ---- (4) dW[co][ci][3][3] = G^T (sum_z dU[z]) G -------------------------------------------------------------------------
// One thread per (ci, co); a workgroup owns 8 ci x 32 co: dU is read in 128-B segments along co, the nine taps go through LDS
// and leave as 288-B runs (8 ci x 9 taps are contiguous in OIHW) instead of 4-B stores scattered Cin*36 B apart.
template <int MO>
__global__ void __launch_bounds__(256) wgrad_finish_kernel(const float* __restrict__ dU, int splits, float* __restrict__ dw,
                                                           int Cin, int Cout) {
    constexpr int AL = MO + 2;
    constexpr int FCI = 8, FCO = 32;
    __shared__ float tile[FCO][FCI * 9 + 1];
    const size_t per = (size_t)Cin * Cout;
    const size_t zstride = (size_t)AL * AL * per;
    const int tilesCo = (Cout + FCO - 1) / FCO;
    const int co0 = (int)(blockIdx.x % tilesCo) * FCO, ci0 = (int)(blockIdx.x / tilesCo) * FCI;
    const int col = threadIdx.x & 31, cil = threadIdx.x >> 5;
    const int co = co0 + col, ci = ci0 + cil;
    if (co < Cout && ci < Cin) {
        const size_t i = (size_t)ci * Cout + co;
        typedef typename WinoWeightT<MO>::type WT;      // (m = 8: G^T dU G in fp64, rounded to fp32 once)
        WT h[3][AL];                       // h[a][nu] = sum_xi G[xi][a] u[xi][nu]
#pragma unroll
        for (int nu = 0; nu < AL; ++nu) {
            WT colv[AL], g3[3];
#pragma unroll
            for (int xi = 0; xi < AL; ++xi) {
                const float* p = dU + (size_t)(xi * AL + nu) * per + i;
                float sum = 0.f;
                for (int z = 0; z < splits; ++z) sum += p[z * zstride];          // fixed order: deterministic
                colv[xi] = sum;
            }
            gt_transform<MO>(colv, g3);
            h[0][nu] = g3[0]; h[1][nu] = g3[1]; h[2][nu] = g3[2];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            WT g3[3];
            gt_transform<MO>(h[r], g3);
            tile[col][cil * 9 + 3 * r + 0] = (float)g3[0];
            tile[col][cil * 9 + 3 * r + 1] = (float)g3[1];
            tile[col][cil * 9 + 3 * r + 2] = (float)g3[2];
        }
    }
    __syncthreads();
    const int nci = min(FCI, Cin - ci0);
    for (int e = threadIdx.x; e < FCO * FCI * 9; e += 256) {
        const int c = e / (FCI * 9), off = e % (FCI * 9);
        if (co0 + c < Cout && off < nci * 9) dw[((size_t)(co0 + c) * Cin + ci0) * 9 + off] = tile[c][off];
    }
}

// The same sums and transforms with one thread per (ci, 4 co, nu): eight times the threads of the kernel above and 16-byte loads.  A
// 128 x 128 layer gave that kernel 64 workgroups with one 4-byte load chain per thread (0.13 ms for 30 MB of partial sums); this one
// puts ~4 MB in flight.  Phase 1: column sums over the K splits (fixed order) + G^T along xi -> LDS; phase 2: G^T along nu; the nine
// taps of 4 ci leave as 144-byte runs.  Bit-identical to wgrad_finish_kernel (same operation order per (ci, co)).
// The first `bias_blocks` workgroups instead sum the columns of dm11 [T][Cout] (the fp32 plane (1, 1) of dM = the tile sums of dY)
// into db: 32 columns x 32 row lanes per workgroup, fp64 per-thread sums, lanes combined in a fixed order -- the bias gradient
// without its own zeroing / summing / converting launches (bbdm_colsum_f32: three per layer, 25 us of a 64-launch-bound step).
template <int MO>
__global__ void __launch_bounds__(256) wgrad_finish_nu_kernel(const float* __restrict__ dU, int splits, float* __restrict__ dw,
                                                              int Cin, int Cout, const float* __restrict__ dm11, int T,
                                                              float* __restrict__ db, int bias_blocks) {
    constexpr int AL = MO + 2;
    constexpr int NUS = AL <= 8 ? 8 : 16;               // nu slots of the thread layout (m = 8: ten of sixteen used, 2 ci per workgroup)
    constexpr int FCI = 32 / NUS, FCO = 32;
    typedef typename WinoWeightT<MO>::type WT;          // (m = 8: G^T dU G in fp64, rounded to fp32 once)
    __shared__ WT hbuf[FCI][FCO][3][AL + 1];
    __shared__ float tile[FCO][FCI * 9 + 1];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < bias_blocks) {
        __shared__ double red[32][8][4];                          // [row lane][column quad][4]
        const int q = tid & 7, rl = tid >> 3, c = (int)blockIdx.x * 32 + 4 * q;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (c < Cout) {
            const float* p = dm11 + c;
            int r = rl;
            for (; r + 224 < T; r += 256) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(r + 32 * u) * Cout);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
            }
            for (; r < T; r += 32) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)r * Cout);
                s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
            }
        }
        red[rl][q][0] = s0; red[rl][q][1] = s1; red[rl][q][2] = s2; red[rl][q][3] = s3;
        __syncthreads();
        if (tid < 32 && (int)blockIdx.x * 32 + tid < Cout) {
            double t = 0.0;
            for (int k = 0; k < 32; ++k) t += red[k][tid >> 2][tid & 3];
            db[blockIdx.x * 32 + tid] = (float)t;
        }
        return;
    }
    const size_t per = (size_t)Cin * Cout;
    const size_t zstride = (size_t)AL * AL * per;
    const int tilesCo = (Cout + FCO - 1) / FCO;
    const int bid = (int)blockIdx.x - bias_blocks;
    const int co0 = (bid % tilesCo) * FCO, ci0 = (bid / tilesCo) * FCI;
    const int q = tid & 7, nu = (tid >> 3) & (NUS - 1), cil = tid / (8 * NUS);
    if (nu < AL && co0 + 4 * q < Cout && ci0 + cil < Cin) {
        const size_t i = (size_t)(ci0 + cil) * Cout + co0 + 4 * q;
        float4 colv[AL];
#pragma unroll
        for (int xi = 0; xi < AL; ++xi) {
            const float* p = dU + (size_t)(xi * AL + nu) * per + i;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int z = 0; z < splits; ++z) {
                const float4 v = *reinterpret_cast<const float4*>(p + z * zstride);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            colv[xi] = sum;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            WT cv[AL], g3[3];
#pragma unroll
            for (int xi = 0; xi < AL; ++xi) cv[xi] = e == 0 ? colv[xi].x : e == 1 ? colv[xi].y : e == 2 ? colv[xi].z : colv[xi].w;
            gt_transform<MO>(cv, g3);
            hbuf[cil][4 * q + e][0][nu] = g3[0];
            hbuf[cil][4 * q + e][1][nu] = g3[1];
            hbuf[cil][4 * q + e][2][nu] = g3[2];
        }
    }
    __syncthreads();
    for (int e = tid; e < FCI * FCO * 3; e += 256) {
        const int c2 = e / (FCO * 3), rem = e - c2 * (FCO * 3), col = rem / 3, r = rem - col * 3;
        WT hv[AL], g3[3];
#pragma unroll
        for (int k = 0; k < AL; ++k) hv[k] = hbuf[c2][col][r][k];
        gt_transform<MO>(hv, g3);
        tile[col][c2 * 9 + 3 * r + 0] = (float)g3[0];
        tile[col][c2 * 9 + 3 * r + 1] = (float)g3[1];
        tile[col][c2 * 9 + 3 * r + 2] = (float)g3[2];
    }
    __syncthreads();
    const int nci = min(FCI, Cin - ci0);
    for (int e = tid; e < FCO * FCI * 9; e += 256) {
        const int c = e / (FCI * 9), off = e % (FCI * 9);
        if (co0 + c < Cout && off < nci * 9) dw[((size_t)(co0 + c) * Cin + ci0) * 9 + off] = tile[c][off];
    }
}

struct TnGeom {
    int wm, tilesM, tilesN, splits, k_per_split;
};
TnGeom tn_geom(int batch, long long K, int M, int N) {
    TnGeom g;
    g.wm = (M % 256 == 0) ? 4 : 2;
    g.tilesM = cdiv(M, g.wm * 64);
    g.tilesN = cdiv(N, TBN);
    const long long base = (long long)batch * g.tilesM * g.tilesN;
    const int target = 768, minK = 512;       // (swept in round 3, tools/wgrad1x1_bench.py: the best of 384 ... 1536 / 256 ... 2048)
    long long splits = (target + base - 1) / base;              // >= 3 workgroups per CU when K allows
    const long long max_splits = K / minK > 1 ? K / minK : 1;   // >= 32 stages per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long kps = (K + splits - 1) / splits;
    kps = (kps + TKC - 1) / TKC * TKC;
    g.k_per_split = (int)kps;
    g.splits = (int)((K + kps - 1) / kps);
    return g;
}

}  // namespace

extern "C" int bbdm_gemm_tn_splits(int batch, long long K, int M, int N) {
    if (batch <= 0 || K <= 0 || M <= 0 || N <= 0) return 0;
    return tn_geom(batch, K, M, N).splits;
}

// C[z][b][M][N] (z < bbdm_gemm_tn_splits: partial sums over disjoint K ranges, to be added in order by the consumer)
int bbdm_gemm_tn_impl(const float* A, int lda, size_t a_stride, const float* B, int ldb, size_t b_stride, float* C, float* bias_part,
                      int batch, long long K, int M, int N, void* stream);
extern "C" int bbdm_gemm_tn_batched_f32(const float* A, int lda, size_t a_stride, const float* B, int ldb, size_t b_stride,
                                        float* C, int batch, long long K, int M, int N, void* stream) {
    return bbdm_gemm_tn_impl(A, lda, a_stride, B, ldb, b_stride, C, nullptr, batch, K, M, N, stream);
}
// ... + bias_part[z][b][N] = the column sums of B over split z's K range (conv_wgrad.hip: the bias gradient of the 1x1 layers)
int bbdm_gemm_tn_impl(const float* A, int lda, size_t a_stride, const float* B, int ldb, size_t b_stride, float* C, float* bias_part,
                      int batch, long long K, int M, int N, void* stream) {
    BBDM_REQUIRE(A && B && C, "gemm_tn: null pointer");
    BBDM_REQUIRE(batch > 0 && K > 0 && M > 0 && N > 0 && M % 4 == 0 && N % 4 == 0, "gemm_tn: bad shape batch=%d K=%lld M=%d N=%d",
                 batch, K, M, N);
    BBDM_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= M && ldb >= N && a_stride % 4 == 0 && b_stride % 4 == 0 &&
                 (((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm_tn: pitch / alignment (lda=%d ldb=%d)", lda, ldb);
    const TnGeom g = tn_geom(batch, K, M, N);
    TnArgs a;
    a.A = A; a.B = B; a.C = C; a.a_stride = a_stride; a.b_stride = b_stride; a.lda = lda; a.ldb = ldb;
    a.M = M; a.N = N; a.K = K; a.k_per_split = g.k_per_split; a.batch = batch; a.splits = g.splits;
    a.tilesM = g.tilesM; a.tilesN = g.tilesN; a.items = batch * g.splits; a.bias_part = bias_part;
    BBDM_REQUIRE(!bias_part || ((uintptr_t)bias_part & 15) == 0, "gemm_tn: bias_part alignment");
    const long long blocks = (long long)cdiv(a.items, 8) * 8 * g.tilesM * g.tilesN;
    BBDM_REQUIRE(blocks < (1ll << 31), "gemm_tn: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (g.wm == 4) {
        constexpr size_t lds = (size_t)2 * TKC * (256 + 32 + TBN + 32) * sizeof(float);
        hipLaunchKernelGGL((gemm_tn_f32<4>), dim3((unsigned)blocks), dim3(512), lds, st, a);
    } else {
        constexpr size_t lds = (size_t)2 * TKC * (128 + 32 + TBN + 32) * sizeof(float);
        hipLaunchKernelGGL((gemm_tn_f32<2>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    }
    BBDM_CHECK_LAUNCH("gemm_tn");
    return BBDM_OK;
}

extern "C" int bbdm_winograd_dy_transform_f32(int m, const float* dy, int ld, float* dM, int N, int H, int W, int Cout,
                                              void* stream) {
    BBDM_WINO_M(m);
    BBDM_REQUIRE(dy && dM && N > 0, "winograd_dy: null pointer / bad N");
    BBDM_WINO_HW(m, H, W);
    BBDM_REQUIRE(Cout > 0 && Cout % 4 == 0 && ld % 4 == 0 && ld >= Cout && (((uintptr_t)dy | (uintptr_t)dM) & 15) == 0,
                 "winograd_dy: Cout=%d ld=%d / 16-byte alignment", Cout, ld);
    const size_t T = wino_tiles_raw(N, H, W, m), Tp = wino_tiles_padded(N, H, W, m);
    const size_t plane = Tp * (size_t)Cout;
    const long long units = (long long)T * (Cout / (m == 6 ? 2 : 4));
    long long blocks = (units + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)blocks), b(256);
    if (m == 2) hipLaunchKernelGGL((winograd_dy_kernel<2, float4>), g, b, 0, st, dy, ld, dM, N, H, W, Cout, plane);
    else if (m == 4) hipLaunchKernelGGL((winograd_dy_kernel<4, float4>), g, b, 0, st, dy, ld, dM, N, H, W, Cout, plane);
    else hipLaunchKernelGGL((winograd_dy_kernel<6, float2>), g, b, 0, st, dy, ld, dM, N, H, W, Cout, plane);
    BBDM_CHECK_LAUNCH("winograd_dy");
    return BBDM_OK;
}

static int wgrad_finish_launch(int m, const float* dU, int splits, float* dw_oihw, int Cin, int Cout, const float* dm11, int T,
                               float* dbias, void* stream) {
    BBDM_WINO_M8(m);
    BBDM_REQUIRE(dU && dw_oihw && splits > 0 && Cin > 0 && Cout > 0, "winograd_wgrad_finish: bad args");
    hipStream_t st = (hipStream_t)stream;
    const dim3 b(256);
    if (Cout % 4 == 0 && ((uintptr_t)dU & 15) == 0 && (!dbias || ((uintptr_t)dm11 & 15) == 0)) {
        const int bias_blocks = dbias ? cdiv(Cout, 32) : 0;
        const long long blocks = (long long)cdiv(Cout, 32) * cdiv(Cin, m == 8 ? 2 : 4) + bias_blocks;     // (FCI of wgrad_finish_nu_kernel)
        BBDM_REQUIRE(blocks < (1ll << 31), "winograd_wgrad_finish: grid too large");
        const dim3 g((unsigned)blocks);
        if (m == 2) hipLaunchKernelGGL((wgrad_finish_nu_kernel<2>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout, dm11, T, dbias, bias_blocks);
        else if (m == 4) hipLaunchKernelGGL((wgrad_finish_nu_kernel<4>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout, dm11, T, dbias, bias_blocks);
        else if (m == 8) hipLaunchKernelGGL((wgrad_finish_nu_kernel<8>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout, dm11, T, dbias, bias_blocks);
        else hipLaunchKernelGGL((wgrad_finish_nu_kernel<6>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout, dm11, T, dbias, bias_blocks);
        BBDM_CHECK_LAUNCH("winograd_wgrad_finish");
        return BBDM_OK;
    }
    BBDM_REQUIRE(!dbias, "winograd_wgrad_finish_bias: Cout %% 4 == 0 and 16-byte aligned dU / dm11 required");
    const long long blocks = (long long)cdiv(Cout, 32) * cdiv(Cin, 8);
    BBDM_REQUIRE(blocks < (1ll << 31), "winograd_wgrad_finish: grid too large");
    const dim3 g((unsigned)blocks);
    if (m == 2) hipLaunchKernelGGL((wgrad_finish_kernel<2>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout);
    else if (m == 4) hipLaunchKernelGGL((wgrad_finish_kernel<4>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout);
    else if (m == 8) hipLaunchKernelGGL((wgrad_finish_kernel<8>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout);
    else hipLaunchKernelGGL((wgrad_finish_kernel<6>), g, b, 0, st, dU, splits, dw_oihw, Cin, Cout);
    BBDM_CHECK_LAUNCH("winograd_wgrad_finish");
    return BBDM_OK;
}

extern "C" int bbdm_winograd_wgrad_finish_f32(int m, const float* dU, int splits, float* dw_oihw, int Cin, int Cout,
                                              void* stream) {
    return wgrad_finish_launch(m, dU, splits, dw_oihw, Cin, Cout, nullptr, 0, nullptr, stream);
}

// ... and the bias gradient in the same launch: db[c] = sum_t dm11[t][c], t < T (dm11: the plane (1, 1) of dM, pitch Cout, as
// bbdm_winograd_dy_transform_bf3p_f32 writes it -- the tile sums of dY).  Cout % 4 == 0, dU / dm11 16-byte aligned.
extern "C" int bbdm_winograd_wgrad_finish_bias_f32(int m, const float* dU, int splits, float* dw_oihw, int Cin, int Cout,
                                                   const float* dm11, long long T, float* dbias, void* stream) {
    BBDM_REQUIRE(dm11 && dbias && T > 0 && T < (1ll << 31), "winograd_wgrad_finish_bias: bad dm11 / dbias / T");
    return wgrad_finish_launch(m, dU, splits, dw_oihw, Cin, Cout, dm11, (int)T, dbias, stream);
}

// workspace of bbdm_conv3x3_winograd_wgrad_f32, in floats: V | dM | dU[splits] | fp64 column-sum scratch [Cout]
extern "C" size_t bbdm_winograd_wgrad_workspace_floats(int m, int N, int H, int W, int Cin, int Cout) {
    if ((m != 2 && m != 4 && m != 6) || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const size_t P = wino_planes(m), Tp = wino_tiles_padded(N, H, W, m);
    const TnGeom g = tn_geom((int)P, (long long)wino_tiles_raw(N, H, W, m), Cin, Cout);
    return P * Tp * ((size_t)Cin + Cout) + (size_t)g.splits * P * Cin * Cout + 8 * (size_t)Cout + 2;      // (+ the colsum's limb cells)
}

// dW (OIHW, overwritten) and, when dbias != NULL, db of y = conv3x3(x, w) + b from x [N,H,W,Cin] (pitch ldx) and
// dy [N,H,W,Cout] (pitch ldy): stages (1)-(4) above + the column sums.  Same result as bbdm_conv_wgrad_f32(ks = 3) up to the
// fp32 rounding quoted at the top of this file.
extern "C" int bbdm_conv3x3_winograd_wgrad_f32(int m, const float* x, int ldx, const float* dy, int ldy, float* dw_oihw,
                                               float* dbias, float* ws, int N, int H, int W, int Cin, int Cout, void* stream) {
    BBDM_WINO_M(m);
    BBDM_REQUIRE(x && dy && dw_oihw && ws, "winograd_wgrad: null pointer");
    BBDM_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0,
                 "winograd_wgrad: bad shape N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
    BBDM_REQUIRE(((uintptr_t)ws & 15) == 0, "winograd_wgrad: workspace alignment");
    const size_t P = wino_planes(m), T = wino_tiles_raw(N, H, W, m), Tp = wino_tiles_padded(N, H, W, m);
    BBDM_REQUIRE(Tp * (size_t)(Cin > Cout ? Cin : Cout) < (1ull << 32), "winograd_wgrad: one transformed plane exceeds 2^32 elements");
    float* V = ws;
    float* dM = V + P * Tp * Cin;
    float* dU = dM + P * Tp * Cout;
    const int splits = bbdm_gemm_tn_splits((int)P, (long long)T, Cin, Cout);
    int rc = bbdm_winograd_input_f32(m, x, ldx, V, nullptr, nullptr, 0, 0, 0, N, H, W, Cin, stream);
    if (rc == BBDM_OK) rc = bbdm_winograd_dy_transform_f32(m, dy, ldy, dM, N, H, W, Cout, stream);
    if (rc == BBDM_OK)
        rc = bbdm_gemm_tn_batched_f32(V, Cin, Tp * (size_t)Cin, dM, Cout, Tp * (size_t)Cout, dU, (int)P, (long long)T, Cin, Cout,
                                      stream);
    if (rc == BBDM_OK) rc = bbdm_winograd_wgrad_finish_f32(m, dU, splits, dw_oihw, Cin, Cout, stream);
    if (rc == BBDM_OK && dbias) {
        // Row 1 of A is all ones (the transform point x = 1), so dM at xi = (1, 1) is the SUM of the tile's dY: the bias
        // gradient is the column sum of that one plane over the tiles -- m^2 x fewer rows than a pass over dY itself.
        size_t off = (size_t)(dU - ws) + (size_t)splits * P * Cin * Cout;
        off = (off + 1) & ~(size_t)1;                                       // 8-byte alignment of the fp64 scratch
        rc = bbdm_colsum_f32(dM + (size_t)(m + 3) * Tp * Cout, Cout, reinterpret_cast<double*>(ws + off), dbias, (long long)T,
                             Cout, stream);
    }
    return rc;
}
