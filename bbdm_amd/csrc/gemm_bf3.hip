// gemm_bf3.hip -- the batched tile GEMMs of the Winograd path, fp32-accurate, on the BF16 matrix core.
//
// CDNA4 runs v_mfma_f32_32x32x2_f32 at the fp32 VECTOR rate (157 TFLOP/s): 1/16 of v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s;
// MI355X_MICROARCH.md).  An fp32 number splits EXACTLY into three bf16 numbers,
//     x = x1 + x2 + x3,   x1 = bf16(x),  x2 = bf16(x - x1),  x3 = bf16(x - x1 - x2)          (3 x 8 significand bits = 24),
// (round-to-nearest at every step; the subtractions are exact in fp32), so a product of two fp32 numbers is
//     a b = a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2  + (a2 b3 + a3 b2 + a3 b3)
// where the bracket is <= 2^-23 |a b| -- the size of one fp32 rounding -- and is dropped.  Every kept term is a bf16 x bf16
// product (exact in fp32) accumulated in fp32 by the matrix core, exactly as the f32 MFMA accumulates: the result is an
// fp32-accurate dot product ("BF16x3 / 6-term" emulation, the scheme vendor BLAS libraries ship as fp32-emulation mode), NOT
// a reduced-precision one -- tests/test_kernels_gpu.py::test_gemm_bf3_* measure it against an fp64 GEMM next to the f32-MFMA
// kernel.  Cost: 6 MFMAs of 32 cycles per 32x32x16 block instead of 8 of 64 cycles -> 2.67x the matrix throughput.
//
//   M_xi[tiles x Cout] = V_xi[tiles x Cin] . U_xi[Cin x Cout]      for the (m+2)^2 transform points xi (gridDim.z)
//
//  * V stays fp32 in HBM (the input transform is untouched); it is split while it is staged: global -> registers ->
//    3 bf16 planes in LDS (5.5 VALU ops per element, on the staging threads, under the MFMAs of the previous phase).
//  * U is split once per weight update into [xi][chunk][3][CoutPad][16] bf16 (bbdm_gemm_bf3_pack_f32).
//  * Workgroup = 8 waves, tile 256 tiles x 128 couts, K walked in chunks of 16: one MFMA K-step.  LDS per stage: 3 x 256 x 32 B
//    + 3 x 128 x 32 B = 36 KB, double buffered.  Rows are 32 B (16 bf16) unpadded; 16-B half h of row r lives at half
//    h ^ ((r >> 3) & 1), so the 16 lanes of a ds_read_b128 group (16 consecutive rows, same half) hit 16 distinct 16-B
//    slots of the 256-B bank window.
//  * Lane l supplies row (l & 31) and the 8 consecutive k of half (l >> 5) for A and for B alike (the sum over k does not
//    depend on which k a lane carries as long as A and B agree); C/D layout = the f32 32x32 layout (cdna_hip_programming.md).
#include "bf3_split.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 128, KC = 16, NTHR = 512;
constexpr int ROWB = 32;                       // bytes per LDS row (16 bf16)
constexpr int A_PLANE = BM * ROWB, W_PLANE = BN * ROWB;
constexpr int STAGE = 3 * A_PLANE + 3 * W_PLANE;      // 36864 B

struct Bf3Args {
    const float* V;          // [batch][T][Cin] fp32
    const unsigned short* U; // [batch][nchunks][3][CoutPad][16] bf16
    float* M;                // [batch][T][Cout] fp32
    size_t vz, uz, mz;       // per-batch strides (elements of the respective type)
    int T, Cin, Cout, CoutPad, nchunks, tilesN;
    int tiles, batch, by_batch;   // by_batch: each XCD owns whole batch entries (1-D launch), see the kernel
    int lda, ldo, ldr;       // row pitches of V, M and the residual (floats)
    const float* bias;       // [Cout] or null
    const float* res;        // [T][ldr] or null; may alias M
};

// byte offset inside a plane: 32-B rows, the 16-B half flips with bits 2 and 3 of the row, so that the 8 (and 16) consecutive
// rows a ds_read_b128 group touches land on distinct 16-B slots of a 128-B (256-B) window.  Measured: indistinguishable from the
// `(row >> 3) & 1` swizzle this kernel first shipped with (tools/bf3_probe.py, +-1 %) -- fragment reads are not what bounds it.
__device__ __forceinline__ int swz(int row, int byte_in_row) {
    return row * ROWB + (byte_in_row ^ ((((row >> 2) ^ (row >> 3)) & 1) << 4));
}

__device__ __forceinline__ int xcd_block(int nblk, int x, int off) {
    if (nblk < 64) return x;
    const int c = (off + x) & 7;
    int start = 0;
    for (int cc = 0; cc < 8; ++cc) {
        if (cc == c) break;
        const int first = (cc - off) & 7;
        start += (nblk - first + 7) >> 3;
    }
    return start + ((x - ((c - off) & 7)) >> 3);
}

// The fp32 A tile of chunk c + 1 is requested from HBM before the MFMAs of chunk c and split + stored after them; a second
// register set that requests it two chunks ahead was measured 8 % SLOWER (15 spilled VGPRs at the 128-register budget of
// two workgroups per CU; round-2 A/B: C2 step 166.0 vs 153.8 ms) and removed; so was a K = 32-per-phase variant with ONE
// workgroup per CU, 227 VGPRs and a 2-deep A ring (tile GEMMs 137.6 vs 154.5 "TFLOP/s", C2 step 149.0 vs 137.0 ms): two
// co-resident workgroups covering each other's barriers matter more than the extra latency tolerance.
// RES: add a residual row in the epilogue (compile-time: a run-time branch per store would serialise the stores behind
// vmcnt(0), see conv_igemm.hip).
template <bool RES>
__global__ void __launch_bounds__(NTHR, 4) gemm_bf3_kernel(const Bf3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                                    // 4 (M) x 2 (N) waves of 64 x 64
    // Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).
    int bid, bz;
    if (a.by_batch) {
        // batched launches (the (m+2)^2 Winograd GEMMs): XCD c owns the batch entries c, c + 8, ... and runs ALL their tiles, so an
        // entry's V rows and U slice are fetched over the fabric by one L2 only (with the row tiles of every entry spread over
        // the 8 XCDs each L2 fetched the whole of U and the Cout tiles' shared V rows arrived 2.4x: 3.8 GB per launch against
        // 2.3 GB algorithmic)
        const int L = (int)blockIdx.x, j = L >> 3;
        bz = (L & 7) + 8 * (j / a.tiles);
        if (bz >= a.batch) return;
        bid = j % a.tiles;
    } else {
        // one GEMM (1x1 layers) or a batch that is no multiple of 8: a contiguous range of the (n_tile fastest) tile order per
        // XCD, so that the Cout tiles of one row tile share its V rows through one L2 (see conv_igemm.hip)
        bz = (int)blockIdx.z;
        bid = xcd_block((int)gridDim.x, (int)blockIdx.x, (int)(((size_t)blockIdx.z * gridDim.x) % 8));
    }
    const float* V = a.V + (size_t)bz * a.vz;
    const unsigned short* U = a.U + (size_t)bz * a.uz;
    float* M = a.M + (size_t)bz * a.mz;
    const int n_tile = bid % a.tilesN, m_tile = bid / a.tilesN;
    const int row0 = m_tile * BM, cout0 = n_tile * BN;

    // ---- staging maps -----------------------------------------------------------------------------------------------------
    // A: 256 rows x 16 fp32 = 1024 float4: thread owns f = tid and tid + 512 -> (row = f >> 2, 4-k group q = f & 3)
    const float* asrc[2];
    int adst[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int f = tid + s * NTHR, row = f >> 2, q = f & 3;
        asrc[s] = V + (size_t)(row0 + row) * a.lda + q * 4;
        adst[s] = swz(row, q * 8);
    }
    // W: per plane 128 rows x 32 B = 512 units of 8 B: thread owns unit tid of each of the 3 planes
    const int wrow = tid >> 2, wq = tid & 3;
    const unsigned short* wsrc = U + ((size_t)cout0 + wrow) * KC + wq * 4;      // + (chunk * 3 + plane) * CoutPad * 16
    const int wdst = swz(wrow, wq * 8);
    const size_t wplane = (size_t)a.CoutPad * KC;

    // ---- fragment addresses -------------------------------------------------------------------------------------------------
    int aoff[2], boff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        aoff[t] = swz(wm * 64 + t * 32 + (lane & 31), (lane >> 5) * 16);
        boff[t] = 3 * A_PLANE + swz(wn * 64 + t * 32 + (lane & 31), (lane >> 5) * 16);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 areg[2];
    uint2 wreg[3];
    auto load = [&](int chunk) {
#pragma unroll
        for (int s = 0; s < 2; ++s) areg[s] = *reinterpret_cast<const float4*>(asrc[s] + chunk * KC);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            wreg[p] = *reinterpret_cast<const uint2*>(wsrc + (size_t)(chunk * 3 + p) * wplane);
    };
    auto store = [&](unsigned char* st) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint2 p1, p2, p3;
            split4(areg[s], p1, p2, p3);
            *reinterpret_cast<uint2*>(st + adst[s]) = p1;
            *reinterpret_cast<uint2*>(st + A_PLANE + adst[s]) = p2;
            *reinterpret_cast<uint2*>(st + 2 * A_PLANE + adst[s]) = p3;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(st + 3 * A_PLANE + p * W_PLANE + wdst) = wreg[p];
    };

    load(0);
    store(smem);
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const bool more = chunk + 1 < a.nchunks;
        if (more) load(chunk + 1);
        const unsigned char* st = smem + (chunk & 1) * STAGE;
        bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[t][p] = *reinterpret_cast<const bf16x8*>(st + p * A_PLANE + aoff[t]);
                bf[t][p] = *reinterpret_cast<const bf16x8*>(st + p * W_PLANE + boff[t]);
            }
        // term-major, tile-minor: consecutive MFMAs go to DIFFERENT accumulators (a dependent MFMA would wait for its
        // predecessor's result); smallest terms first: they meet an accumulator not yet grown by this chunk's leading term
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][BF3_TA[t]], bf[j][BF3_TB[t]], acc[i][j], 0, 0, 0);
        if (more) store(smem + ((chunk + 1) & 1) * STAGE);
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual); 32 lanes x 4 B = one 128-B line per store instruction ------------------------------
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = cout0 + wn * 64 + j * 32 + (lane & 31);
        bv[j] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            float rv[8][2];
            if (RES) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = r0 + rr;
                    const int row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int co = cout0 + wn * 64 + j * 32 + (lane & 31);
                        rv[rr][j] = co < a.Cout ? a.res[(size_t)row * a.ldr + co] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = r0 + rr;
                const int row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float* dst = M + (size_t)row * a.ldo;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co = cout0 + wn * 64 + j * 32 + (lane & 31);
                    float v = acc[i][j][r] + bv[j];
                    if (RES) v += rv[rr][j];
                    if (co < a.Cout) dst[co] = v;
                }
            }
        }
}

// fp32 packed [batch][nchunks][CoutPad][16] (the layout conv_igemm.hip's 1x1 mode takes) -> [batch][nchunks][3][CoutPad][16] bf16
__global__ void bf3_pack_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t per_batch_chunks,
                                int CoutPad) {
    const size_t slab = (size_t)CoutPad * KC;                  // elements per (batch, chunk)
    const size_t total = per_batch_chunks * slab;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = i / slab, e = i - bc * slab;
        const float x = src[i];
        const float x1 = bf16_round(x);
        const float r1 = x - x1;
        const float x2 = bf16_round(r1);
        const float x3 = bf16_round(r1 - x2);
        unsigned short* d = dst + bc * 3 * slab + e;
        d[0] = (unsigned short)(__float_as_uint(x1) >> 16);
        d[slab] = (unsigned short)(__float_as_uint(x2) >> 16);
        d[2 * slab] = (unsigned short)(__float_as_uint(x3) >> 16);
    }
}

}  // namespace

extern "C" size_t bbdm_gemm_bf3_packed_halfs(int batch, int CinPad, int Cout) {
    return (size_t)batch * cdiv(CinPad, KC) * 3 * (cdiv(Cout, 128) * 128) * KC;
}

extern "C" int bbdm_gemm_bf3_pack_f32(const float* packed_f32, void* packed_bf3, int batch, int CinPad, int Cout, void* stream) {
    BBDM_REQUIRE(packed_f32 && packed_bf3 && batch > 0 && CinPad > 0 && Cout > 0, "gemm_bf3_pack: bad args");
    const int CoutPad = cdiv(Cout, 128) * 128;
    const size_t chunks = (size_t)batch * cdiv(CinPad, KC);
    const size_t total = chunks * CoutPad * KC;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bf3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed_f32,
                       (unsigned short*)packed_bf3, chunks, CoutPad);
    BBDM_CHECK_LAUNCH("gemm_bf3_pack");
    return BBDM_OK;
}

// Can this shape take the bf16x3 kernel?  (whole 256-row tiles, whole 16-channel chunks)
extern "C" int bbdm_gemm_bf3_supported(long long T, int CinPad, int Cout) {
    return T > 0 && T % BM == 0 && CinPad > 0 && CinPad % KC == 0 && Cout > 0 && Cout % 4 == 0 &&
           (size_t)T * (size_t)CinPad < (1ull << 32);
}

template <bool RES>
static int bf3_launch(const Bf3Args& a, long long blocks, int batch, hipStream_t st) {
    static bool attr_set_dev[BBDM_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[bbdm_device_slot()];
    const size_t lds = 2 * STAGE;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<RES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            bbdm_set_error("gemm_bf3: hipFuncSetAttribute(%zu B LDS) failed", lds);
            return BBDM_E_LAUNCH;
        }
        attr_set = true;
    }
    if (a.by_batch)
        hipLaunchKernelGGL((gemm_bf3_kernel<RES>), dim3((unsigned)(8 * blocks * ((batch + 7) / 8))), dim3(NTHR), lds, st, a);
    else
        hipLaunchKernelGGL((gemm_bf3_kernel<RES>), dim3((unsigned)blocks, 1, batch), dim3(NTHR), lds, st, a);
    return BBDM_OK;
}

static int bf3_run(Bf3Args& a, int batch, hipStream_t st) {
    a.CoutPad = cdiv(a.Cout, 128) * 128;
    a.nchunks = a.Cin / KC;
    a.tilesN = a.CoutPad / BN;
    a.uz = (size_t)a.nchunks * 3 * a.CoutPad * KC;
    const long long blocks = ((long long)a.T / BM) * a.tilesN;
    BBDM_REQUIRE(blocks * ((batch + 7) / 8) * 8 < (1ll << 31), "gemm_bf3: too many tiles");
    const int by_batch_env = 1;       // batched launches: each XCD owns whole batch entries (multiples of 8)
    a.tiles = (int)blocks;
    a.batch = batch;
    a.by_batch = (by_batch_env && batch >= 8 && (batch % 8 == 0 || by_batch_env == 2)) ? 1 : 0;
    const int rc = a.res ? bf3_launch<true>(a, blocks, batch, st) : bf3_launch<false>(a, blocks, batch, st);
    if (rc != BBDM_OK) return rc;
    BBDM_CHECK_LAUNCH("gemm_bf3");
    return BBDM_OK;
}

extern "C" int bbdm_gemm_bf3_f32(const float* V, const void* packed_bf3, float* M, int batch, long long T, int CinPad, int Cout,
                                 void* stream) {
    BBDM_REQUIRE(V && packed_bf3 && M && batch > 0, "gemm_bf3: null pointer / bad batch");
    BBDM_REQUIRE(bbdm_gemm_bf3_supported(T, CinPad, Cout), "gemm_bf3: T=%lld CinPad=%d Cout=%d unsupported (T %% 256, CinPad %% 16)",
                 T, CinPad, Cout);
    BBDM_REQUIRE((((uintptr_t)V | (uintptr_t)M | (uintptr_t)packed_bf3) & 15) == 0, "gemm_bf3: 16-byte alignment");
    Bf3Args a;
    a.V = V; a.U = (const unsigned short*)packed_bf3; a.M = M;
    a.T = (int)T; a.Cin = CinPad; a.Cout = Cout;
    a.lda = CinPad; a.ldo = Cout; a.ldr = 0; a.bias = nullptr; a.res = nullptr;
    a.vz = (size_t)T * CinPad; a.mz = (size_t)T * Cout;
    return bf3_run(a, batch, (hipStream_t)stream);
}

// 1x1 convolution / Linear on NHWC activations = one GEMM [pixels x CinPad] . [CinPad x Cout] + bias (+ residual), with the
// operand pitches of channel slices (the UNet's skip 1x1 convs, qkv / proj_out, the SpatialTransformer's Linears).
// packed_bf3: bbdm_gemm_bf3_pack_f32(batch = 1) of the buffer bbdm_conv_pack_weight_f32(ks = 1) filled.
extern "C" int bbdm_conv1x1_bf3_f32(const float* x, int ldx, const void* packed_bf3, const float* bias, const float* residual,
                                    int ldr, float* out, int ldo, long long pixels, int CinPad, int Cout, void* stream) {
    BBDM_REQUIRE(x && packed_bf3 && out, "conv1x1_bf3: null pointer");
    BBDM_REQUIRE(bbdm_gemm_bf3_supported(pixels, CinPad, Cout), "conv1x1_bf3: pixels=%lld CinPad=%d Cout=%d unsupported", pixels,
                 CinPad, Cout);
    BBDM_REQUIRE(ldx % 4 == 0 && ldx >= CinPad && ldo >= Cout && (!residual || ldr >= Cout) &&
                     (((uintptr_t)x | (uintptr_t)packed_bf3) & 15) == 0,
                 "conv1x1_bf3: bad pitch / alignment");
    BBDM_REQUIRE((size_t)pixels * (size_t)ldx < (1ull << 32), "conv1x1_bf3: input exceeds 2^32 elements");
    Bf3Args a;
    a.V = x; a.U = (const unsigned short*)packed_bf3; a.M = out;
    a.T = (int)pixels; a.Cin = CinPad; a.Cout = Cout;
    a.lda = ldx; a.ldo = ldo; a.ldr = ldr; a.bias = bias; a.res = residual;
    a.vz = 0; a.mz = 0;
    return bf3_run(a, 1, (hipStream_t)stream);
}
