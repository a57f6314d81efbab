// gemm_bf3p.hip -- the fp32-accurate bf16x3 GEMM of gemm_bf3.hip with BOTH operands pre-split by their producers.
//
// gemm_bf3.hip takes fp32 rows and splits them into their three bf16 planes while it stages them: global -> registers -> 11 VALU
// ops per pair -> three ds_writes, on the same waves that issue the MFMAs.  Its ablation (profiles/r02_bf3_ablation.txt: MFMAs
// alone 2.38 ms, staging alone 1.68 ms, together 3.38 ms) says that work overlaps the matrix pipe only half.  Here the A operand
// arrives ALREADY split -- the Winograd input transform (winograd.hip: winograd_input_split_kernel) or bbdm_gemm_bf3p_split_rows_f32
// write the three planes -- and already in the order the matrix core wants it, so the GEMM's main loop is LDS-DMA copies,
// fragment reads and MFMAs: no VALU arithmetic, no ds_write, no staging registers.  Arithmetic is gemm_bf3.hip's to the bit (the
// same exact three-way split, the same six terms in the same order, fp32 accumulation in the MFMA).
//
// Operand layout ("fragment units"): a unit is 32 rows x 16 k of ONE bf16 plane = 1 KB, stored in the order the lanes of
// v_mfma_f32_32x32x16_bf16 consume it: lane l supplies row (l & 31) and the 8 consecutive k of half (l >> 5), so element (r, k)
// sits at byte (k >> 3) * 512 + r * 16 + (k & 7) * 2 and a wave reads its fragment with ONE ds_read_b128 at lane * 16 -- linear,
// bank-conflict free by construction, and exactly the image one global_load_lds_dwordx4 (wave-uniform LDS base + lane * 16)
// deposits when every lane fetches global unit + lane * 16: one perfectly coalesced 1 KB request per unit, no swizzle anywhere.
//   A (V planes): [batch][T / 32 row groups][nchunks][3 planes][1 KB]      6 B per element (fp32: 4 B)
//   B (U planes): [batch][CoutPad / 32 col groups][nchunks][3 planes][1 KB]
// The chunks of one row group are contiguous: a workgroup streams its 8 row groups as 8 sequential 3 KB-per-chunk streams.
//
// Kernel: workgroup = WM x WN waves of 64 x 64 outputs (2 x 2 MFMA tiles, 64 accumulator VGPRs), K walked 16 at a time (one MFMA
// K-step): per chunk and wave 12 fragment reads, 24 MFMAs and its share of the next chunks' unit copies into the other LDS stage.
// gemm_bf3p_pipe_kernel: the fragments of chunk c + 1 are read and the copies of chunk c + 2 issued BETWEEN the MFMAs of chunk c, so
// that after the barrier every wave continues with MFMAs at once; 256 x 256 tiles (16 waves, one workgroup per CU: a third less
// L2 -> LDS traffic per FLOP than 256 x 128) where Cout fills them.
// Measured (MI355X, the 42 Winograd layers of the C2 step): 216 TFLOP/s fp32-equivalent = 1.30 PFLOP/s of bf16 MFMA = 0.52 of the
// 416.7 (2500 / 6) peak, MfmaUtil 75 % at the 1.77 GHz the chip sustains under this load; gemm_bf3.hip: 181 = 0.43, MfmaUtil 54 %.
// Also in this file (same layouts, same bits): gemm_bf3q_pipe_kernel -- the pipelined kernel with an fp32 A operand split by the waves
// between their MFMAs (wide 1x1 layers); gemm_bf3s_kernel -- the small-problem 1x1 kernel (64 x 64 tiles, 64 channels per step,
// both operands by LDS-DMA two steps ahead, fp32 A split at the fragment read: round 4); the NS = 3 build of the pipe kernel for
// small launches whose workgroups have a CU to themselves (round 4).
#include "bf3_split.h"
#include "h2_split.h"
#include "lds_dma.h"
#include <stdlib.h>

namespace {

constexpr int KC = 16;
constexpr int UNIT = 1024;                     // bytes of one fragment unit

struct Bf3pArgs {
    const unsigned char* A;  // [batch][T / 32][nchunks][3][UNIT]
    const unsigned char* B;  // [batch][CoutPad / 32][nchunks][3][UNIT]
    float* M;                // [batch][T][ldo] fp32
    size_t az, bz, mz, rz;   // per-batch strides: bytes, bytes, floats (M), floats (residual)
    int T, Cout, nchunks, tilesN;
    int rgs;                 // pipe kernel: 32-row groups the A planes are LAID OUT for (>= T / 32; the groups beyond T hold the producer's zeros)
    int tiles, batch, by_batch;
    int persist;             // pipe kernel, by_batch: workgroups walk the tiles blockIdx.x + k gridDim.x
    int ksplits, kps, P;     // split-K (pipe kernel): batch index = z * P + entry; split z walks chunks [z kps, (z+1) kps) of nchunks
    int ldo, ldr;
    const float* bias;       // [Cout] or null
    const float* res;        // [T][ldr] or null; may alias M
    // h2 planes (NP = 2: two fp16 planes under a power-of-two scale, h2_split.h): the bounds the two operands were scaled by -- the
    // epilogue multiplies the accumulators by 2^-(eA + eB), exact -- or null (bf16x3 planes carry no scale)
    const float* hA = nullptr;
    const float* hB = nullptr;
    float gA = 1.f;          // *hA bounds the tensor the A planes were FORMED from; gA = the gain of that transform (Winograd: wino_input_gain)
    float gB = 1.f;          // ... likewise for B (the weight gradient's dM = A dY A^T under the bound of dY: wino_dy_gain)
};

__device__ __forceinline__ int xcd_block_p(int nblk, int x, int off) {
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

typedef int frag_t __attribute__((ext_vector_type(4)));       // one lane's 8 bf16 of an MFMA operand, as the registers it occupies
#define BF3P_BF(x) __builtin_bit_cast(bf16x8, x)

// LDS byte address of a __shared__ object (the operand of a hand-written ds_read)
__device__ __forceinline__ unsigned lds_address(void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)p; }

// ---- software-pipelined variant: the fragments of chunk c + 1 are read WHILE the MFMAs of chunk c run ------------------------------
// (The plain two-stage structure -- reads, copy issue, MFMAs, ONE wait + barrier per chunk -- measured 181 - 204 TFLOP/s against this
// kernel's 211 - 216 on the C2 layers, profiles/r03_bf3p_variants.txt, and is no longer built.)  There every wave did its front-end work (fragment reads, copy issue) right after the barrier -- all waves at
// once, the matrix pipe idle meanwhile (measured: ~12 % of the kernel with one 16-wave workgroup per CU).  Here a wave enters an
// iteration with the fragments of its chunk already in registers and issues, between the six term groups of its 24 MFMAs, the
// copies of chunk c + 2 and the reads of chunk c + 1 -- each fragment register is re-loaded right after the last term that uses
// it (term order BF3_TA / BF3_TB = (1,1) (0,2) (2,0) (0,1) (1,0) (0,0): plane 2 of B dies first, then plane 2 of A, plane 1 of B,
// ... -- bit-equal results), so no second register set is needed.  Two LDS stages: chunk c + 2 goes to
// the stage chunk c was read from (one iteration ago).  One wait (copies landed, reads returned) + barrier per chunk, and after
// it every wave continues with MFMAs at once.
// Persistent tiles (a.persist: by_batch launches with more tiles than the chip holds workgroups): a workgroup walks the tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... -- gridDim.x is a multiple of 8, so a tile stays on the XCD the one-tile-per-workgroup
// launch would have given it -- and issues the first two chunk copies of its NEXT tile before it stores the current tile's
// accumulators: the workgroup turnover (drain the stores, free 96 KB of LDS, launch, first copy latency: ~7 us of a ~145 us
// K = 1024 tile, from the K = 1024 / K = 2048 timings) shrinks to the store issue.
// NS = LDS stages (round 4).  Two serve the large problems: sixteen waves per CU and operands that mostly come from L2 keep the matrix
// pipe fed although every iteration ends with vmcnt(0), i.e. a chunk's copies get ONE chunk of MFMAs (~0.4 us) to land.  With NS > 2
// chunk c + NS is requested during chunk c and the iteration only waits for chunk c + 2 (counted vmcnt: the NS - 2 younger chunks stay
// in flight; vmcnt retires in order).  The launcher (bf3p_forward) uses it for SMALL problems whose workgroups all fit the chip at once
// with the larger LDS footprint.  Register budget: the ring's addressing does not fit the 128 VGPRs of the four-waves-per-SIMD build
// (it spilled ~200 registers, and scratch traffic shares vmcnt with the copies: wrong results on hardware) -- NS > 2 is compiled for
// the occupancy its LDS footprint allows anyway (NS = 3: two 4-wave workgroups per CU, deeper: one).  The same builds showed two codegen
// traps that cost more than the ring gained until they were removed: fragments held as <8 x bf16> across the has_next branches are
// legalised element-wise (120 v_perm / v_lshrrev per iteration), and with the branches kept the register allocator copies the whole
// fragment set around each of them (108 v_mov_b64 per iteration) -- hence frag_t and the unconditional reads below.
// NP = planes per operand (round 6): 3 = the bf16x3 split, six terms; 2 = the fp16 pair of h2_split.h, three terms on
// v_mfma_f32_32x32x16_f16 and the accumulators re-scaled by 2^-(eA + eB) in the epilogue.  Same layout with NP units per chunk.
// R4 (round 6, with NP = 2): a ring of FOUR chunk slots and TWO chunks per barrier.  With three terms instead of six a chunk is 12 MFMAs
// per wave, so one wait + barrier per chunk weighed twice as much as on the bf16x3 planes (the first h2 build ran its MFMAs at 0.37 -
// 0.45 of the peak against the bf16x3 kernel's 0.47 - 0.54).  Here an iteration owns chunks c, c + 1: it enters with the fragments of c in
// registers, reads those of c + 1 (landed and visible since the last barrier: no barrier in between) under the terms of c and those
// of c + 2 under the terms of c + 1, and issues at its top the copies of chunks c + 3 / c + 4 into the slots of c - 1 / c, which every wave
// has finished reading before the last barrier.  A chunk's copies get a whole iteration (24 MFMAs per wave) to land, as on bf16x3.
template <int WM, int WN, bool RES, int NS = 2, int NP = 3, bool R4 = false>
__global__ void __launch_bounds__(WM * WN * 64, NS == 2 ? 4 : NS == 3 ? 2 : 1) gemm_bf3p_pipe_kernel(const Bf3pArgs a) {
    static_assert(!R4 || (NP == 2 && NS == 2), "the four-slot ring is the fp16-pair kernel's");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [NS][STAGE]  (R4: [4][STAGE])
    constexpr int NW = WM * WN, BM = WM * 64, BN = WN * 64;
    constexpr int NA = WM * 2 * NP, NB = WN * 2 * NP, NU = NA + NB, STAGE = NU * UNIT;
    constexpr int KMAX = (NU + NW - 1) / NW;
    static_assert(NS == 2 || NU % NW == 0, "counted waits need the same number of copies per wave and chunk");
    static_assert(NS >= 2 && (NS - 2) * KMAX <= 60, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned lane16 = lane * 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tilesN = a.tilesN * 2 / WN;
    const size_t gstride = (size_t)a.nchunks * NP * UNIT;
    // A ragged last row tile (T % BM != 0) reads the row groups the buffer holds beyond the real rows -- zeros, written by the producer
    // of the planes (Winograd input transform, split pass) -- and past THEM the last group again; its rows are not stored.  (Round 5:
    // it used to re-read the last REAL group: the idle 32-row blocks then multiplied live data, and the kernel is bound by the power its
    // operand data draws -- zeros run 1.3x faster)
    const int rg_last = a.rgs - 1;
    // ---- the tile a (virtual) block index names; false beyond the last batch entry ----------------------------------------------
    int row0 = 0, cout0 = 0, n = 0;
    float* M = nullptr;
    const float* res = nullptr;
    const unsigned char* src[KMAX];
    auto setup = [&](int L) -> bool {
        int bid, bz;
        if (a.by_batch) {
            // XCD x = L % 8 owns the batch entries x, x + 8, ...: an entry's B planes cross the fabric into ONE L2.  A last group of 1, 2
            // or 4 entries (batch % 8: the 36 points of F(4x4,3x3), the 100 of F(8x8,3x3)) is SHARED instead of padded -- entry
            // 8 g + x % rem, every (8 / rem)-th of its tiles per XCD -- so that every XCD gets batch / 8 entries' worth of tiles (round 5:
            // with whole entries four XCDs of an F(4x4) layer held 5 entries and four 4, and the layer took the time of 5: +11 %)
            const int j = L >> 3, x = L & 7, g = j / a.tiles;
            const int full = a.batch >> 3, rem = a.batch & 7;
            if (g < full) {
                bz = x + 8 * g;
                bid = j - g * a.tiles;
            } else if (rem == 1 || rem == 2 || rem == 4) {
                const int t = (j - full * a.tiles) * (8 / rem) + x / rem;
                if (t >= a.tiles) return false;
                bz = 8 * full + x % rem;
                bid = t;
            } else {
                bz = x + 8 * g;
                if (bz >= a.batch) return false;
                bid = j - g * a.tiles;
            }
        } else {
            bz = (int)blockIdx.z;
            bid = xcd_block_p((int)gridDim.x, (int)blockIdx.x, (int)(((size_t)blockIdx.z * gridDim.x) % 8));
        }
        const int n_tile = bid % tilesN, m_tile = bid / tilesN;
        row0 = m_tile * BM; cout0 = n_tile * BN;
        // split-K (the weight-gradient GEMMs: few output tiles, long contraction): entry = bz % P, split z = bz / P
        const int zs = a.ksplits > 1 ? bz / a.P : 0, ent = a.ksplits > 1 ? bz - zs * a.P : bz;
        const size_t koff = (size_t)zs * a.kps * (NP * UNIT);
        const unsigned char* A = a.A + (size_t)ent * a.az + koff;
        const unsigned char* B = a.B + (size_t)ent * a.bz + koff + (size_t)n_tile * (WN * 2) * gstride;
        M = a.M + (size_t)bz * a.mz;
        res = RES ? a.res + (size_t)bz * a.rz : nullptr;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int u = wave + k * NW;
            const int ub = u - NA;
            src[k] = (u < NA ? A + (size_t)min(m_tile * (WM * 2) + u / NP, rg_last) * gstride + (u % NP) * UNIT
                             : B + (size_t)(ub / NP) * gstride + (ub % NP) * UNIT);      // wave-uniform (SGPRs); + lane * 16 below
        }
        n = a.ksplits > 1 ? min(a.kps, a.nchunks - zs * a.kps) : a.nchunks;
        return true;
    };
    auto issue = [&](int chunk, unsigned char* st) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int u = wave + k * NW;
            if ((k + 1) * NW <= NU || u < NU) glds16(src[k] + (size_t)chunk * (NP * UNIT), lane16, st + u * UNIT);
        }
    };
    auto load_bias = [&](float (&bv)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = cout0 + wn * 64 + j * 32 + (lane & 31);
            bv[j] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        }
    };
    const unsigned lds0 = lds_address(smem);
    const unsigned aoff = (wm * 2) * NP * UNIT + lane * 16;
    const unsigned boff = (NA + (wn * 2) * NP) * UNIT + lane * 16;

    int L = (int)blockIdx.x;
    if (!setup(L)) return;
    // fp16-pair planes: 2^-(eA + eB), read once at the top (a wave-uniform value in an SGPR; before any hand-written LDS read is in flight)
    float descale = 1.f;
    if constexpr (NP == 2) {
        const float d = h2_pow2(-(h2_exp_of_bound(*a.hA * a.gA) + h2_exp_of_bound(*a.hB * a.gB)));
        descale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(d)));
    }
    float bv[2];
    load_bias(bv);
#pragma unroll
    for (int k = 0; k < (R4 ? 3 : NS); ++k)
        if (k < n) issue(k, smem + k * STAGE);

    // (held as 4 x 32-bit: a <8 x bf16> value that lives across the has_next branches is legalised ELEMENT-wise by the compiler --
    // 120 v_lshrrev / v_perm per iteration in the NS > 2 builds; a bit-cast at the MFMA is free)
    frag_t fa[NP][2], fb[NP][2];                                               // [plane][tile]
#define BF3P_READ(dst, base, p, t) do { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(((t) * NP + (p)) * UNIT)); } while (0)
#define BF3P_READ_A(p, base) do { BF3P_READ(fa[p][0], base, p, 0); BF3P_READ(fa[p][1], base, p, 1); } while (0)
#define BF3P_READ_B(p, base) do { BF3P_READ(fb[p][0], base, p, 0); BF3P_READ(fb[p][1], base, p, 1); } while (0)
#define BF3P_READS_RETURNED()                                                                                                  \
    do {                                                                                                                        \
        if constexpr (NP == 3)                                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[NP - 1][0]), "+v"(fa[NP - 1][1]),  \
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[NP - 1][0]), "+v"(fb[NP - 1][1])   \
                         :: "memory");                                                                                          \
        else                                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]),                                      \
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])                                       \
                         :: "memory");                                                                                          \
    } while (0)
// the fp16-pair ring's counted waits (LDS returns in order: lgkmcnt(4) = everything but the four youngest reads, A 1 and B 0)
#define BF3P_FIRST_READS_RETURNED()                                                                                            \
    do { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]) :: "memory"); } while (0)
#define BF3P_LAST_READS_RETURNED()                                                                                             \
    do { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[0][0]), "+v"(fb[0][1]) :: "memory"); } while (0)
#define BF3P_ALL_LANDED()                                                                                                      \
    do {                                                                                                                        \
        wait_vmcnt<0>();                                                                                                        \
        BF3P_READS_RETURNED();                                                                                                  \
    } while (0)
#define BF3P_MFMA(A_, B_, C_)                                                                                                   \
    (NP == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF3P_BF(A_), BF3P_BF(B_), C_, 0, 0, 0)                                    \
             : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0))
#define BF3P_TERM(pa, pb)                                                                                                       \
    do {                                                                                                                        \
        acc[0][0] = BF3P_MFMA(fa[pa][0], fb[pb][0], acc[0][0]);                                                                 \
        acc[0][1] = BF3P_MFMA(fa[pa][0], fb[pb][1], acc[0][1]);                                                                 \
        acc[1][0] = BF3P_MFMA(fa[pa][1], fb[pb][0], acc[1][0]);                                                                 \
        acc[1][1] = BF3P_MFMA(fa[pa][1], fb[pb][1], acc[1][1]);                                                                 \
    } while (0)
    for (;;) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        wait_vmcnt<0>();
        asm volatile("" :: "v"(bv[0]), "v"(bv[1]));  // a USE of the bias: hipcc waits for its load here, where it can see the wait (it does not see the
                                                     // waits inside the asm statements below and would re-wait vmcnt(0) before every epilogue store)
        asm volatile("s_barrier" ::: "memory");
        {
            const unsigned sa = lds0 + aoff, sb = lds0 + boff;
#pragma unroll
            for (int p = 0; p < NP; ++p) { BF3P_READ_A(p, sa); BF3P_READ_B(p, sb); }
        }
        BF3P_ALL_LANDED();
        asm volatile("s_barrier" ::: "memory");                                // everybody holds chunk 0: stage 0 may be overwritten
        if constexpr (R4) {
            for (int chunk = 0; chunk < n; chunk += 2) {
                const bool has1 = chunk + 1 < n, has2 = chunk + 2 < n;
                // term order (h1 k2) (h1 k1) (h2 k1): the first term of a chunk uses A plane 0 and B plane 1, whose last use in the previous
                // chunk was its first / second term -- their reads were issued one or two term groups (4 - 8 MFMAs per wave) earlier, so the
                // counted wait in the middle of the iteration finds them returned; the planes the LAST term frees (A 1, B 0) are read last
                // and waited for after the first term of the next chunk.  (With (h1 k2) (h2 k1) (h1 k1) the first term needed A plane 0,
                // freed by the last: every wave -- the four of a SIMD run in step after a barrier -- sat out an LDS round trip per chunk.)
                {   // ---- chunk: its fragments are in registers; read chunk + 1 (landed since the last barrier) ----------------------
                    const unsigned nxt = lds0 + ((chunk + 1) & 3) * STAGE;
                    const unsigned sa = nxt + aoff, sb = nxt + boff;
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_TERM(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (chunk + 3 < n) issue(chunk + 3, smem + ((chunk + 3) & 3) * STAGE);     // the slot of chunk - 1
                    if (chunk + 4 < n) issue(chunk + 4, smem + (chunk & 3) * STAGE);           // the slot of chunk (in registers)
                    if (has1) BF3P_READ_B(1, sb);
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_TERM(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1) BF3P_READ_A(0, sa);
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_TERM(1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1) { BF3P_READ_A(1, sa); BF3P_READ_B(0, sb); }
                }
                if (has1) {   // ---- chunk + 1; read chunk + 2 -----------------------------------------------------------------------
                    BF3P_FIRST_READS_RETURNED();                               // B 1 and A 0 (the four youngest reads may be in flight)
                    const unsigned nxt = lds0 + ((chunk + 2) & 3) * STAGE;
                    const unsigned sa = nxt + aoff, sb = nxt + boff;
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_TERM(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_LAST_READS_RETURNED();                                // A 1 and B 0, requested a term group ago
                    if (has2) BF3P_READ_B(1, sb);
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_TERM(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (has2) BF3P_READ_A(0, sa);
                    __builtin_amdgcn_sched_barrier(0);
                    BF3P_TERM(1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (has2) { BF3P_READ_A(1, sa); BF3P_READ_B(0, sb); }
                }
                wait_vmcnt<0>();
                BF3P_READS_RETURNED();
                asm volatile("s_barrier" ::: "memory");
            }
        } else
        for (int chunk = 0; chunk < n; ++chunk) {
            // (NS > 2: the reads are unconditional -- in the last iteration they fetch a stale stage into registers nobody uses; with the
            // branches the compiler, given the larger register budget of those builds, copied the whole fragment set around each of them:
            // 108 v_mov_b64 per iteration)
            const bool has_next = NS > 2 || chunk + 1 < n;
            const unsigned nxt = lds0 + ((chunk + 1) % NS) * STAGE;
            const unsigned sa = nxt + aoff, sb = nxt + boff;
            if constexpr (NP == 3) {
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(1, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (chunk + NS < n) issue(chunk + NS, smem + (chunk % NS) * STAGE);   // under the first MFMAs; stage free since the last barrier
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(0, 2);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) BF3P_READ_B(2, sb);
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(2, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) BF3P_READ_A(2, sa);
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) BF3P_READ_B(1, sb);
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) BF3P_READ_A(1, sa);
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) { BF3P_READ_A(0, sa); BF3P_READ_B(0, sb); }
            } else {
                // fp16 pair: (h1 k2) (h1 k1) (h2 k1) -- the order of the ring kernel above (results are bit-equal across the tile variants);
                // each fragment is re-read right after its last use
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (chunk + NS < n) issue(chunk + NS, smem + (chunk % NS) * STAGE);
                if (has_next) BF3P_READ_B(1, sb);
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) BF3P_READ_A(0, sa);
                __builtin_amdgcn_sched_barrier(0);
                BF3P_TERM(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) { BF3P_READ_A(1, sa); BF3P_READ_B(0, sb); }
            }
            // my copies of chunk + 2 have landed (NS > 2: those of chunks chunk + 3 .. chunk + NS, issued after them, may stay in flight;
            // in the last iterations fewer are behind them: wait for all), my reads of chunk + 1 returned
            if (NS > 2 && chunk + NS < n) wait_vmcnt<(NS - 2) * KMAX>();
            else wait_vmcnt<0>();
            BF3P_READS_RETURNED();
            asm volatile("s_barrier" ::: "memory");                            // ... everybody's
        }
        // ---- the next tile's first copies go out before this tile's stores (both LDS stages are free: the last barrier is behind) --
        const int crow0 = row0, ccout0 = cout0;
        float* const cM = M;
        const float* const cres = res;
        const float cb0 = bv[0], cb1 = bv[1];
        L += (int)gridDim.x;
        const bool more = a.persist && setup(L);
        if (more) {
#pragma unroll
            for (int k = 0; k < (R4 ? 3 : NS); ++k)
                if (k < n) issue(k, smem + k * STAGE);
            load_bias(bv);
        }
        // ---- epilogue: + bias (+ residual); 32 lanes x 4 B = one 128-B line per store instruction ------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float rv[8][2];
                if (RES) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = r0 + rr;
                        const int row = min(crow0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), a.T - 1);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int co = ccout0 + wn * 64 + j * 32 + (lane & 31);
                            rv[rr][j] = co < a.Cout ? cres[(size_t)row * a.ldr + co] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = r0 + rr;
                    const int row = crow0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float* dst = cM + (size_t)row * a.ldo;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int co = ccout0 + wn * 64 + j * 32 + (lane & 31);
                        float v = (NP == 2 ? acc[i][j][r] * descale : acc[i][j][r]) + (j ? cb1 : cb0);
                        if (RES) v += rv[rr][j];
#if BBDM_NT_MSTORE
                        if (co < a.Cout && row < a.T) store_nt(dst + co, v);
#else
                        if (co < a.Cout && row < a.T) dst[co] = v;
#endif
                    }
                }
            }
        if (!more) break;
    }
#undef BF3P_TERM
#undef BF3P_MFMA
#undef BF3P_ALL_LANDED
#undef BF3P_LAST_READS_RETURNED
#undef BF3P_FIRST_READS_RETURNED
#undef BF3P_READS_RETURNED
#undef BF3P_READ_B
#undef BF3P_READ_A
#undef BF3P_READ
}

// ---- the pipelined kernel with an fp32 A operand: every wave splits ITS share of the next-but-one chunk between its MFMAs -------------
// gemm_bf3p_pipe_kernel needs A as planes (6 B per element, written by a producer that knows it); gemm_bf3.hip takes plain fp32 rows
// and splits them while it stages -- on 8-wave workgroups that do load / MFMA / split + store / barrier in lock step (181 TFLOP/s).
// This kernel keeps the pipe kernel's structure -- fragments of chunk c + 1 read and chunk c + 2 staged BETWEEN the MFMAs of chunk c,
// B planes by LDS-DMA, 256-column tiles, one 16-wave workgroup per CU -- and stages A through registers: a thread requests its 16 B
// of chunk c + 2 at the top of iteration c and, after the fourth term group, splits them (bf3_split.h: 22 VALU instructions) and
// writes 3 x 8 B into the stage chunk c was read from.  One float4 + 22 VALU instructions per wave and chunk against its 24 MFMAs:
// half of gemm_bf3.hip's split work per MFMA, and no wave waits for another's split.
// A = fp32 rows with a pitch (the 1x1 convolutions: x NHWC).  (Round 3 also fed it the Winograd V as fp32 row units, 4 B per element
// instead of the planes' 6: input transforms -3.7 ms, tile GEMMs +3.9 ms on the C2 step, profiles/r03_bf3q_bench.txt; removed.)
// Results bit-equal to the other bf16x3 kernels.
// Register budget: the pipe kernel's 64 accumulators + 48 fragment registers leave no room for the staging registers and the split's
// temporaries at 128 VGPRs (a 16-wave build spilled 194 registers and ran at 20 TFLOP/s): the workgroup is 3 x WN waves -- 192-row
// tiles, THREE waves per SIMD, up to 168 VGPRs each (WN = 4: one 12-wave workgroup per CU; WN = 2: two 6-wave workgroups).
// NP = 2 (round 6): the fp16-pair planes of h2_split.h -- A scaled by 2^eA (eA from the bound *a.hA of the activation) and split into two
// fp16 halves (7 VALU instructions per pair instead of 11), B planes packed under their own bound, three f16 MFMA terms, the accumulators
// re-scaled in the epilogue.  32 fragment registers instead of 48: the workgroup is 4 x WN waves (256-row tiles) at 128 VGPRs.
template <int WM, int WN, bool RES, int NP = 3>
__global__ void __launch_bounds__(WM * WN * 64, NP == 2 && WM == 4 ? 4 : 3) gemm_bf3q_pipe_kernel(const Bf3pArgs a, const float* __restrict__ Af, int lda) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][STAGE]
    constexpr int NW = WM * WN, BM = WM * 64, BN = WN * 64;
    static_assert((BM * 4) % (NW * 64) == 0, "whole float4 slots of A per thread and chunk");
    constexpr int NA = WM * 2 * NP, NB = WN * 2 * NP, NU = NA + NB, STAGE = NU * UNIT;
    constexpr int KB = (NB + NW - 1) / NW;                                    // B copies per wave and chunk
    constexpr int AS = (BM * 4 + NW * 64 - 1) / (NW * 64);                    // float4 of A per thread and chunk (1 at 16 waves, 2 at 8)
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned lane16 = lane * 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tilesN = a.tilesN * 2 / WN;
    const size_t gstride = (size_t)a.nchunks * NP * UNIT;
    int bid, bz;
    if (a.by_batch) {
        const int L = (int)blockIdx.x, j = L >> 3;
        bz = (L & 7) + 8 * (j / a.tiles);
        if (bz >= a.batch) return;
        bid = j % a.tiles;
    } else {
        bz = (int)blockIdx.z;
        bid = xcd_block_p((int)gridDim.x, (int)blockIdx.x, (int)(((size_t)blockIdx.z * gridDim.x) % 8));
    }
    const int n_tile = bid % tilesN, m_tile = bid / tilesN;
    const int row0 = m_tile * BM, cout0 = n_tile * BN;
    const int n = a.nchunks;
    float* M = a.M + (size_t)bz * a.mz;
    const float* res = RES ? a.res + (size_t)bz * a.rz : nullptr;
    // ---- A staging: slot f = tid (+ NW * 64): row f >> 2 of the tile, k-quad q = f & 3 (k = 4 q .. 4 q + 3) -------------------------------
    const float* asrc[AS];
    unsigned adst[AS];
    size_t astep;                                                             // floats between consecutive chunks of a slot
#pragma unroll
    for (int s = 0; s < AS; ++s) {
        const int f = tid + s * NW * 64, row = f >> 2, q = f & 3;
        const int grow = min(row0 + row, a.T - 1);                            // (a ragged last row tile re-reads the last row)
        asrc[s] = Af + (size_t)bz * (a.az / 4) + (size_t)grow * lda + q * 4;
        // plane unit of row group (row >> 5): byte (k >> 3) * 512 + r' * 16 + (k & 7) * 2, k = 4 q, with the row slot of the SECOND k-half
        // rotated by four, r' = r ^ 4 (round 5): a 16-lane group of the 8-byte stores covers 4 rows x 4 k-quads, and quads 0 and 2 of a
        // row -- 512 B apart -- fell on the same banks (2-way: SQ_LDS_BANK_CONFLICT 16 % of this kernel's LDS cycles, round-4 counters);
        // now quads 2 / 3 land 16 banks further.  The A fragments are read back through the same rotation (aoff below); a read group's
        // 16 consecutive rows stay a permutation of one aligned 256-B window: conflict-free as before.  (This unit layout is private to
        // the kernel: its own waves write and read it.)
        adst[s] = (unsigned)(((row >> 5) * NP) * UNIT + (q >> 1) * 512 + (((row & 31) ^ ((q >> 1) << 2)) * 16) + (q & 1) * 8);
    }
    astep = KC;
    const unsigned char* bsrc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int u = wave + k * NW;
        bsrc[k] = a.B + (size_t)bz * a.bz + (size_t)n_tile * (WN * 2) * gstride + (size_t)(u / NP) * gstride + (u % NP) * UNIT;
    }
    auto issue_b = [&](int chunk, unsigned char* st) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int u = wave + k * NW;
            if ((k + 1) * NW <= NB || u < NB) glds16(bsrc[k] + (size_t)chunk * (NP * UNIT), lane16, st + (NA + u) * UNIT);
        }
    };
    float4 areg[AS];
    auto load_a = [&](int chunk) {
#pragma unroll
        for (int s = 0; s < AS; ++s) areg[s] = *reinterpret_cast<const float4*>(asrc[s] + (size_t)chunk * astep);
    };
    // fp16-pair planes: the scale of A and the factor the accumulators are re-scaled by (wave-uniform; read before any LDS read is in flight)
    float ascale = 1.f, descale = 1.f;
    if constexpr (NP == 2) {
        const int ea = h2_exp_of_bound(*a.hA * a.gA), eb = h2_exp_of_bound(*a.hB * a.gB);
        ascale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(h2_pow2(ea))));
        descale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(h2_pow2(-(ea + eb)))));
    }
    auto store_a = [&](unsigned char* st) {
#pragma unroll
        for (int s = 0; s < AS; ++s) {
            if constexpr (NP == 2) {
                uint2 p1, p2;
                h2_split4(make_float4(areg[s].x * ascale, areg[s].y * ascale, areg[s].z * ascale, areg[s].w * ascale), p1, p2);
                *reinterpret_cast<uint2*>(st + adst[s]) = p1;
                *reinterpret_cast<uint2*>(st + adst[s] + UNIT) = p2;
            } else {
                uint2 p1, p2, p3;
                split4(areg[s], p1, p2, p3);
                *reinterpret_cast<uint2*>(st + adst[s]) = p1;
                *reinterpret_cast<uint2*>(st + adst[s] + UNIT) = p2;
                *reinterpret_cast<uint2*>(st + adst[s] + 2 * UNIT) = p3;
            }
        }
    };
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = cout0 + wn * 64 + j * 32 + (lane & 31);
        bv[j] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
    const unsigned lds0 = lds_address(smem);
    const unsigned aoff = (wm * 2) * NP * UNIT + (lane >> 5) * 512 + (((lane & 31) ^ ((lane >> 5) << 2)) * 16);      // (the rotation of adst)
    const unsigned boff = (NA + (wn * 2) * NP) * UNIT + lane * 16;
    // prologue: chunks 0 and 1 staged, fragments of chunk 0 in registers
    issue_b(0, smem);
    load_a(0);
    store_a(smem);
    if (n > 1) {
        issue_b(1, smem + STAGE);
        load_a(1);
        store_a(smem + STAGE);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    frag_t fa[NP][2], fb[NP][2];                                               // [plane][tile] (as the registers they occupy)
#define BF3Q_READ(dst, base, p, t) do { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(((t) * NP + (p)) * UNIT)); } while (0)
#define BF3Q_READ_A(p, base) do { BF3Q_READ(fa[p][0], base, p, 0); BF3Q_READ(fa[p][1], base, p, 1); } while (0)
#define BF3Q_READ_B(p, base) do { BF3Q_READ(fb[p][0], base, p, 0); BF3Q_READ(fb[p][1], base, p, 1); } while (0)
#define BF3Q_ALL_LANDED()                                                                                                      \
    do {                                                                                                                        \
        wait_vmcnt<0>();                                                                                                        \
        if constexpr (NP == 3)                                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[NP - 1][0]), "+v"(fa[NP - 1][1]),  \
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[NP - 1][0]), "+v"(fb[NP - 1][1])   \
                         :: "memory");                                                                                          \
        else                                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]),                                      \
                           "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])                                       \
                         :: "memory");                                                                                          \
    } while (0)
#define BF3Q_MFMA(A_, B_, C_)                                                                                                   \
    (NP == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF3P_BF(A_), BF3P_BF(B_), C_, 0, 0, 0)                                    \
             : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0))
#define BF3Q_TERM(pa, pb)                                                                                                       \
    do {                                                                                                                        \
        acc[0][0] = BF3Q_MFMA(fa[pa][0], fb[pb][0], acc[0][0]);                                                                 \
        acc[0][1] = BF3Q_MFMA(fa[pa][0], fb[pb][1], acc[0][1]);                                                                 \
        acc[1][0] = BF3Q_MFMA(fa[pa][1], fb[pb][0], acc[1][0]);                                                                 \
        acc[1][1] = BF3Q_MFMA(fa[pa][1], fb[pb][1], acc[1][1]);                                                                 \
    } while (0)
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(bv[0]), "v"(bv[1]));
    asm volatile("s_barrier" ::: "memory");
    {
        const unsigned sa = lds0 + aoff, sb = lds0 + boff;
#pragma unroll
        for (int p = 0; p < NP; ++p) { BF3Q_READ_A(p, sa); BF3Q_READ_B(p, sb); }
    }
    BF3Q_ALL_LANDED();
    asm volatile("s_barrier" ::: "memory");                                // everybody holds chunk 0: stage 0 may be overwritten
    for (int chunk = 0; chunk < n; ++chunk) {
        const bool has_next = chunk + 1 < n, stage2 = chunk + 2 < n;
        const unsigned nxt = lds0 + ((chunk + 1) & 1) * STAGE;
        const unsigned sa = nxt + aoff, sb = nxt + boff;
        unsigned char* const st2 = smem + (chunk & 1) * STAGE;             // chunk + 2 goes where chunk was read from (one iteration ago)
        if constexpr (NP == 3) {
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (stage2) { issue_b(chunk + 2, st2); load_a(chunk + 2); }        // under the first MFMAs
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(0, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) BF3Q_READ_B(2, sb);
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) BF3Q_READ_A(2, sa);
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) BF3Q_READ_B(1, sb);
            if (stage2) store_a(st2);                                          // (waits for this thread's A request; the split runs under the MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) BF3Q_READ_A(1, sa);
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) { BF3Q_READ_A(0, sa); BF3Q_READ_B(0, sb); }
        } else {
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (stage2) { issue_b(chunk + 2, st2); load_a(chunk + 2); }        // under the first MFMAs
            if (has_next) BF3Q_READ_B(1, sb);
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(0, 0);                                                   // (the term order of gemm_bf3p_pipe_kernel's fp16 pair)
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) BF3Q_READ_A(0, sa);
            if (stage2) store_a(st2);                                          // (waits for this thread's A request; the split runs under the MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            BF3Q_TERM(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) { BF3Q_READ_A(1, sa); BF3Q_READ_B(0, sb); }
        }
        BF3Q_ALL_LANDED();
        asm volatile("s_barrier" ::: "memory");
    }
#undef BF3Q_TERM
#undef BF3Q_MFMA
#undef BF3Q_ALL_LANDED
#undef BF3Q_READ_B
#undef BF3Q_READ_A
#undef BF3Q_READ
    // ---- epilogue: + bias (+ residual); 32 lanes x 4 B = one 128-B line per store instruction ------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            float rv[8][2];
            if (RES) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = r0 + rr;
                    const int row = min(row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), a.T - 1);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int co = cout0 + wn * 64 + j * 32 + (lane & 31);
                        rv[rr][j] = co < a.Cout ? res[(size_t)row * a.ldr + co] : 0.f;
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
                    float v = (NP == 2 ? acc[i][j][r] * descale : acc[i][j][r]) + bv[j];
                    if (RES) v += rv[rr][j];
                    if (co < a.Cout && row < a.T) dst[co] = v;
                }
            }
        }
}

// ---- small 1x1 layers: 64 x 64 tiles, 64 channels per iteration, fp32 A split at the fragment read (round 4) -------------------------
// The 1x1 convolutions of the SMALL configurations (latent / 64^2-pixel models: qkv / proj_out of the attention blocks, the skip
// projections; a few hundred to a few thousand pixels, K up to 2048) ran on the split-K f32-MFMA kernel (conv_igemm.hip): 96 ... 512
// workgroups that each walk a chain of 16-channel chunks, a wait + barrier per chunk, then a second launch that adds the partial sums
// (512 x 1024 -> 3072: 56 us; 1 ms of the 3.4 ms LBBDM-f16 step for all of them).  Such a launch is bound by the LENGTH of that chain
// (an iteration cannot be shorter than a barrier-synchronised LDS round trip plus whatever memory latency it exposes), not by the
// matrix pipe.  This kernel shortens the chain 4x and hides the memory latency inside the workgroup:
//   * one iteration = FOUR chunks (64 channels): K = 1024 is 16 iterations, no split-K, bias / residual in the epilogue, one launch;
//   * 64 x 64 output tiles (4 waves of 32 x 32) so that even 512 x 1024 gives 128 workgroups;
//   * everything arrives by LDS-DMA into a THREE-stage ring, requested two iterations ahead (counted vmcnt waits: no register loads
//     the compiler would guard with vmcnt(0)): B as pre-split planes (bbdm_gemm_bf3p_pack_b_f32: 12 KB contiguous per 32 couts and
//     iteration), A as the fp32 rows lie in HBM (256 B per row and iteration);
//   * A is split when a wave reads its fragment: a lane's 8 channels of one row = two ds_read_b128, 44 VALU instructions (bf3_split.h)
//     that issue between the chunk's six MFMAs.  The raw tile is stored [4 rows][16 pieces of 16 B] per 1 KB copy with the piece index
//     XOR-ed by the row (row % 16): the copy reads whole 256-B row segments from HBM, and the 16 lanes of a fragment read that sit
//     on consecutive rows hit 16 different bank groups.
//   * XCD-aware tile order where Cout gives a multiple of 8 column tiles: XCD c owns the column tiles c, c + 8, ... for every row
//     tile, so a weight byte enters exactly one L2.
// 120 KB of LDS, one workgroup per CU.  K a multiple of 64.  Arithmetic: the six product terms per 16-channel chunk in BF3_TA / BF3_TB
// order, chunks in ascending order -- bit-equal to bbdm_conv1x1_bf3_f32 / bbdm_conv1x1_bf3q_f32.
// NP = 2 (round 6): B as fp16-pair planes (a third fewer weight bytes: these launches are bound by the weight stream), A scaled and split
// into two fp16 halves at the fragment read, three f16 MFMA terms, the accumulator re-scaled in the epilogue (h2_split.h).
template <bool RES, int NS, int NP = 3>
__global__ void __launch_bounds__(256, 1) gemm_bf3s_kernel(const Bf3pArgs a, const float* __restrict__ Af, int lda) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [NS][ASTAGE + BSTAGE]
    constexpr int CH = 4;                                                       // chunks per iteration
    constexpr int ASTAGE = 64 * CH * KC * 4;                                    // 64 rows x 64 channels fp32 = 16 KB
    constexpr int BSTAGE = 2 * CH * NP * UNIT;                                  // 2 row groups x 4 chunks x NP planes x 1 KB = 24 (16) KB
    constexpr int STAGE = ASTAGE + BSTAGE;
    constexpr int KA = ASTAGE / UNIT / 4, KB = BSTAGE / UNIT / 4;               // copies per wave and iteration: 4 of A, 6 of B
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned lane16 = lane * 16;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesM = (a.T + 63) / 64, tilesN = (a.Cout + 63) / 64;
    int m_tile, n_tile;
    if (a.by_batch) {                                                           // (here: "XCD-owned column tiles", see above)
        const int L = (int)blockIdx.x, j = L >> 3;
        n_tile = (L & 7) + 8 * (j / tilesM);
        m_tile = j % tilesM;
    } else {
        m_tile = (int)blockIdx.x % tilesM;
        n_tile = (int)blockIdx.x / tilesM;
    }
    if (n_tile >= tilesN) return;
    const int row0 = m_tile * 64, cout0 = n_tile * 64;
    const int n = a.nchunks / CH;                                               // iterations
    const size_t gstride = (size_t)a.nchunks * NP * UNIT;
    // ---- A copies: copy j = wave + 4 k covers rows 4 j .. 4 j + 3 of the tile; lane = (row in copy, slot), slot holds piece slot ^ (row % 16)
    const unsigned char* asrc[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const int row = (wave + 4 * k) * 4 + (lane >> 4), piece = (lane & 15) ^ (row & 15);
        const int grow = min(row0 + row, a.T - 1);                              // (a ragged last row tile re-reads the last row)
        asrc[k] = reinterpret_cast<const unsigned char*>(Af + (size_t)grow * lda) + piece * 16;
    }
    // ---- B copies: the 24 units of an iteration = 2 row groups x (4 chunks x 3 planes: contiguous in HBM) --------------------------------
    const unsigned char* bsrc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int u = wave + k * 4, g = u / (CH * NP), v = u % (CH * NP);
        bsrc[k] = a.B + (size_t)(n_tile * 2 + g) * gstride + (size_t)v * UNIT;
    }
    auto issue = [&](int it, unsigned char* st) {
#pragma unroll
        for (int k = 0; k < KA; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[k] + (size_t)it * (CH * KC * 4)),
                                             (__attribute__((address_space(3))) void*)(st + (wave + 4 * k) * UNIT), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < KB; ++k) glds16(bsrc[k] + (size_t)it * (CH * NP * UNIT), lane16, st + ASTAGE + (wave + k * 4) * UNIT);
    };
    // ---- fragment reads: lane = (row r = lane % 32 of the wave's 32 rows, half h = lane / 32: channels 8 h .. 8 h + 7 of a chunk) ------
    const int R = wm * 32 + (lane & 31), h = lane >> 5;
    const unsigned abase = (unsigned)((R >> 2) * UNIT + (R & 3) * 256);
    unsigned aoff[CH][2];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        aoff[c][0] = abase + (unsigned)((((4 * c + 2 * h) ^ (R & 15))) * 16);
        aoff[c][1] = abase + (unsigned)((((4 * c + 2 * h + 1) ^ (R & 15))) * 16);
    }
    const unsigned boff = (unsigned)(ASTAGE + (wn * CH * NP) * UNIT) + lane16;
    float ascale = 1.f, descale = 1.f;                                          // fp16 pair: read once, before any LDS read is in flight
    if constexpr (NP == 2) {
        const int ea = h2_exp_of_bound(*a.hA * a.gA), eb = h2_exp_of_bound(*a.hB * a.gB);
        ascale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(h2_pow2(ea))));
        descale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(h2_pow2(-(ea + eb)))));
    }
    float bv = 0.f;
    {
        const int co = cout0 + wn * 32 + (lane & 31);
        bv = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // counted wait: iteration `it` has landed, the (at most NS - 2) younger ones that were requested may stay in flight
    auto landed = [&](int it) {
        const int younger = min(NS - 2, n - 1 - it);
        if (NS >= 4 && younger >= 2) wait_vmcnt<2 * (KA + KB)>();
        else if (younger >= 1) wait_vmcnt<KA + KB>();
        else wait_vmcnt<0>();
    };
#pragma unroll
    for (int k = 0; k < NS - 1; ++k)
        if (k < n) issue(k, smem + k * STAGE);
    landed(0);
    asm volatile("" :: "v"(bv));                                                // (the bias load is older than the copies: it has returned)
    asm volatile("s_barrier" ::: "memory");
    // (hand-written ds_reads, as in the kernels above: the compiler guards a plain LDS load that follows an LDS-DMA copy with
    // vmcnt(0), i.e. it would wait for the copies requested a moment ago.)
    // Software pipeline over the four chunks of an iteration: while the six MFMAs of chunk c issue (32 matrix-pipe cycles each), the
    // wave splits the A fragment of chunk c + 1 (46 VALU instructions: ~8 per MFMA, requested from the scheduler as 1 MFMA : 8 VALU
    // groups) and the reads of chunk c + 2 are in flight -- three register sets.  (Without it MFMAs and splits alternated: ~1.05 us per
    // iteration measured against 0.4 of MFMAs.)
    const unsigned lds0 = lds_address(smem);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ra[3][2];
    frag_t rb[3][NP];
    frag_t fa[2][NP];
#define BF3S_READ(set, c, base)                                                                                                 \
    do {                                                                                                                        \
        asm volatile("ds_read_b128 %0, %1" : "=v"(ra[set][0]) : "v"((base) + aoff[c][0]));                                      \
        asm volatile("ds_read_b128 %0, %1" : "=v"(ra[set][1]) : "v"((base) + aoff[c][1]));                                      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[set][0]) : "v"((base) + boff), "n"(((c) * NP + 0) * UNIT));      \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[set][1]) : "v"((base) + boff), "n"(((c) * NP + 1) * UNIT));      \
        if constexpr (NP == 3)                                                                                                  \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[set][NP - 1]) : "v"((base) + boff), "n"(((c) * NP + NP - 1) * UNIT)); \
    } while (0)
    // the reads of `set` have returned; NEWER = the reads of one younger set (2 + NP) may stay in flight (LDS returns in order)
#define BF3S_RETURNED(set, YOUNGER)                                                                                             \
    do {                                                                                                                        \
        if constexpr (NP == 3) {                                                                                                \
            if (YOUNGER) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0]), "+v"(rb[set][1]), "+v"(rb[set][NP - 1]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0]), "+v"(rb[set][1]), "+v"(rb[set][NP - 1]) :: "memory"); \
        } else {                                                                                                                \
            if (YOUNGER) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0]), "+v"(rb[set][1]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0]), "+v"(rb[set][1]) :: "memory"); \
        }                                                                                                                       \
    } while (0)
#define BF3S_SPLIT(set, dst)                                                                                                    \
    do {                                                                                                                        \
        if constexpr (NP == 3) {                                                                                                \
            uint2 p1, p2, p3, q1, q2, q3;                                                                                       \
            split4(make_float4(ra[set][0][0], ra[set][0][1], ra[set][0][2], ra[set][0][3]), p1, p2, p3);                        \
            split4(make_float4(ra[set][1][0], ra[set][1][1], ra[set][1][2], ra[set][1][3]), q1, q2, q3);                        \
            fa[dst][0] = frag_t{(int)p1.x, (int)p1.y, (int)q1.x, (int)q1.y};                                                    \
            fa[dst][1] = frag_t{(int)p2.x, (int)p2.y, (int)q2.x, (int)q2.y};                                                    \
            fa[dst][NP - 1] = frag_t{(int)p3.x, (int)p3.y, (int)q3.x, (int)q3.y};                                               \
        } else {                                                                                                                \
            uint2 p1, p2, q1, q2;                                                                                               \
            h2_split4(make_float4(ra[set][0][0] * ascale, ra[set][0][1] * ascale, ra[set][0][2] * ascale, ra[set][0][3] * ascale), p1, p2); \
            h2_split4(make_float4(ra[set][1][0] * ascale, ra[set][1][1] * ascale, ra[set][1][2] * ascale, ra[set][1][3] * ascale), q1, q2); \
            fa[dst][0] = frag_t{(int)p1.x, (int)p1.y, (int)q1.x, (int)q1.y};                                                    \
            fa[dst][1] = frag_t{(int)p2.x, (int)p2.y, (int)q2.x, (int)q2.y};                                                    \
        }                                                                                                                       \
    } while (0)
#define BF3S_MFMAS(src, set)                                                                                                    \
    do {                                                                                                                        \
        if constexpr (NP == 3) {                                                                                                \
            _Pragma("unroll") for (int t = 0; t < 6; ++t)                                                                       \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF3P_BF(fa[src][BF3_TA[t] % NP]), BF3P_BF(rb[set][BF3_TB[t] % NP]), acc, 0, 0, 0); \
        } else {                                                                                                                \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[src][0]), __builtin_bit_cast(f16x8, rb[set][1]), acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[src][0]), __builtin_bit_cast(f16x8, rb[set][0]), acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[src][1]), __builtin_bit_cast(f16x8, rb[set][0]), acc, 0, 0, 0); \
        }                                                                                                                       \
    } while (0)
#define BF3S_INTERLEAVE()                                                                                                       \
    do {                                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < (NP == 3 ? 6 : 3); ++t) {                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                                                  \
        }                                                                                                                       \
    } while (0)
    for (int i = 0; i < n; ++i) {
        if (i + NS - 1 < n) issue(i + NS - 1, smem + ((i + NS - 1) % NS) * STAGE);   // the stage iteration i - 1 was read from: free since the barrier
        const unsigned st = lds0 + (unsigned)((i % NS) * STAGE);
        BF3S_READ(0, 0, st);
        BF3S_READ(1, 1, st);
        BF3S_RETURNED(0, 1);
        BF3S_SPLIT(0, 0);
        BF3S_READ(2, 2, st);
        BF3S_RETURNED(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        BF3S_MFMAS(0, 0);                                                       // chunk 0 | split of chunk 1
        BF3S_SPLIT(1, 1);
        BF3S_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        BF3S_READ(0, 3, st);
        BF3S_RETURNED(2, 1);
        __builtin_amdgcn_sched_barrier(0);
        BF3S_MFMAS(1, 1);                                                       // chunk 1 | split of chunk 2
        BF3S_SPLIT(2, 0);
        BF3S_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        BF3S_RETURNED(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        BF3S_MFMAS(0, 2);                                                       // chunk 2 | split of chunk 3
        BF3S_SPLIT(0, 1);
        BF3S_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        BF3S_MFMAS(1, 0);                                                       // chunk 3
        // iteration i + 1 landed (younger requests may stay in flight); everybody is done reading iteration i
        landed(i + 1);
        asm volatile("s_barrier" ::: "memory");
    }
#undef BF3S_INTERLEAVE
#undef BF3S_MFMAS
#undef BF3S_SPLIT
#undef BF3S_RETURNED
#undef BF3S_READ
    // ---- epilogue: + bias (+ residual); 32 lanes x 4 B = one 128-B line per store instruction ------------------------------------------
    const int co = cout0 + wn * 32 + (lane & 31);
    const float* res = RES ? a.res : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < a.Cout && row < a.T) {
            float v = (NP == 2 ? acc[r] * descale : acc[r]) + bv;
            if (RES) v += res[(size_t)row * a.ldr + co];
            a.M[(size_t)row * a.ldo + co] = v;
        }
    }
}

// byte offset of element (row r, k) inside a fragment unit
__device__ __forceinline__ int unit_off(int r, int k) { return (k >> 3) * 512 + r * 16 + (k & 7) * 2; }

// fp32 packed [batch][nchunks][CoutPad][16] (bbdm_conv_pack_weight_f32 ks = 1 / bbdm_winograd_pack_weight_f32) ->
// B planes [batch][CoutPad / 32][nchunks][3][UNIT]; one thread = one (cout, k pair)
__global__ void bf3p_pack_b_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t batch_chunks,
                                   int nchunks, int CoutPad) {
    const size_t pairs = batch_chunks * CoutPad * (KC / 2);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        const int kp = (int)(i % (KC / 2));
        size_t t = i / (KC / 2);
        const int co = (int)(t % CoutPad);
        const size_t bc = t / CoutPad;
        const size_t b = bc / nchunks;
        const int chunk = (int)(bc % nchunks);
        const float2 v = *reinterpret_cast<const float2*>(src + (bc * CoutPad + co) * KC + kp * 2);
        unsigned p1, p2, p3;
        split2(v.x, v.y, p1, p2, p3);
        unsigned char* d = dst + (((b * (CoutPad / 32) + co / 32) * nchunks + chunk) * 3) * (size_t)UNIT + unit_off(co & 31, kp * 2);
        *reinterpret_cast<unsigned*>(d) = p1;
        *reinterpret_cast<unsigned*>(d + UNIT) = p2;
        *reinterpret_cast<unsigned*>(d + 2 * UNIT) = p3;
    }
}

// fp32 rows [batch][T][ld] -> A planes [batch][T / 32][nchunks][3][UNIT].  Workgroup = 32 rows x 8 channel pairs (one unit per
// plane): a wave writes 8 rows x 16 B twice = two full 128-B lines per store instruction.
__global__ void __launch_bounds__(256) bf3p_split_rows_kernel(const float* __restrict__ x, int ld, size_t xz,
                                                               unsigned char* __restrict__ dst, size_t dz, int T, int nchunks) {
    const int chunk = (int)blockIdx.x % nchunks, g = (int)blockIdx.x / nchunks, b = (int)blockIdx.y;
    const int r = threadIdx.x >> 3, kp = threadIdx.x & 7;
    const int row = g * 32 + r;
    float2 v = make_float2(0.f, 0.f);
    if (row < T) v = *reinterpret_cast<const float2*>(x + (size_t)b * xz + (size_t)row * ld + chunk * KC + kp * 2);
    unsigned p1, p2, p3;
    split2(v.x, v.y, p1, p2, p3);
    unsigned char* d = dst + (size_t)b * dz + (((size_t)g * nchunks + chunk) * 3) * UNIT + unit_off(r, kp * 2);
    *reinterpret_cast<unsigned*>(d) = p1;
    *reinterpret_cast<unsigned*>(d + UNIT) = p2;
    *reinterpret_cast<unsigned*>(d + 2 * UNIT) = p3;
}

}  // namespace

extern "C" size_t bbdm_gemm_bf3p_a_bytes(int batch, long long T, int CinPad) {
    return (size_t)batch * (size_t)((T + 255) / 256 * 256) * (size_t)CinPad * 6;
}
extern "C" size_t bbdm_gemm_bf3p_b_bytes(int batch, int CinPad, int Cout) {
    return (size_t)batch * (size_t)(cdiv(Cout, 128) * 128) * (size_t)CinPad * 6;
}

// Can this shape take the pre-split kernel?  (whole 256-row tiles, whole 16-channel chunks)
extern "C" int bbdm_gemm_bf3p_supported(long long T, int CinPad, int Cout) {
    return T > 0 && T % 256 == 0 && CinPad > 0 && CinPad % KC == 0 && Cout > 0 && Cout % 4 == 0 && T < (1ll << 31);
}

extern "C" int bbdm_gemm_bf3p_pack_b_f32(const float* packed_f32, void* b_planes, int batch, int CinPad, int Cout, void* stream) {
    BBDM_REQUIRE(packed_f32 && b_planes && batch > 0 && CinPad > 0 && CinPad % KC == 0 && Cout > 0, "gemm_bf3p_pack_b: bad args");
    const int CoutPad = cdiv(Cout, 128) * 128, nchunks = CinPad / KC;
    const size_t pairs = (size_t)batch * nchunks * CoutPad * (KC / 2);
    size_t blocks = (pairs + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bf3p_pack_b_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed_f32,
                       (unsigned char*)b_planes, (size_t)batch * nchunks, nchunks, CoutPad);
    BBDM_CHECK_LAUNCH("gemm_bf3p_pack_b");
    return BBDM_OK;
}

// fp32 rows -> the A planes (rows T .. the next multiple of 256 are written as zeros).  The Winograd path does not need it (its
// input transform writes the planes); it serves GEMMs whose A operand is an ordinary fp32 matrix, and the tests.
extern "C" int bbdm_gemm_bf3p_split_rows_f32(const float* x, int ldx, void* a_planes, int batch, long long T, int CinPad,
                                             void* stream) {
    BBDM_REQUIRE(x && a_planes && batch > 0 && T > 0 && CinPad > 0 && CinPad % KC == 0 && ldx >= CinPad && ldx % 2 == 0,
                 "gemm_bf3p_split_rows: bad args");
    BBDM_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)a_planes & 15) == 0, "gemm_bf3p_split_rows: alignment");
    const long long Tp = (T + 255) / 256 * 256;
    const int nchunks = CinPad / KC;
    BBDM_REQUIRE((Tp / 32) * nchunks < (1ll << 31) && batch < 65536, "gemm_bf3p_split_rows: too large");
    hipLaunchKernelGGL(bf3p_split_rows_kernel, dim3((unsigned)((Tp / 32) * nchunks), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, (size_t)T * ldx, (unsigned char*)a_planes, (size_t)Tp * CinPad * 6, (int)T,
                       nchunks);
    BBDM_CHECK_LAUNCH("gemm_bf3p_split_rows");
    return BBDM_OK;
}

// tile slots per XCD of a by_batch launch (see setup() in the kernel): whole groups of 8 entries, then the last 1, 2 or 4 entries shared
static inline long long by_batch_slots(long long blocks, int batch) {
    const int full = batch >> 3, rem = batch & 7;
    if (rem == 1 || rem == 2 || rem == 4) return blocks * full + (blocks + 8 / rem - 1) / (8 / rem);
    return blocks * ((batch + 7) / 8);
}
template <int WM, int WN, bool RES, int NS = 2, int NP = 3>
static int bf3p_launch(Bf3pArgs& a, int batch, hipStream_t st) {
    static bool attr_set_dev[BBDM_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[bbdm_device_slot()];
    // the fp16-pair kernel on 256 x 256 tiles: the four-slot ring (one workgroup per CU either way; 128 of the 160 KB)
    constexpr bool R4 = NP == 2 && NS == 2 && WM == 4 && WN == 4;
    const size_t lds = (R4 ? 4 : NS) * (size_t)(WM * 2 * NP + WN * 2 * NP) * UNIT;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3p_pipe_kernel<WM, WN, RES, NS, NP, R4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            bbdm_set_error("gemm_bf3p: hipFuncSetAttribute(%zu B LDS) failed", lds);
            return BBDM_E_LAUNCH;
        }
        attr_set = true;
    }
    const long long blocks = (((long long)a.T + WM * 64 - 1) / (WM * 64)) * (a.tilesN * 2 / WN);     // (the last row tile may be ragged)
    BBDM_REQUIRE(blocks * ((batch + 7) / 8) * 8 < (1ll << 31), "gemm_bf3p: too many tiles");
    a.tiles = (int)blocks;
    dim3 grid = a.by_batch ? dim3((unsigned)(8 * by_batch_slots(blocks, batch))) : dim3((unsigned)blocks, 1, batch);
    a.persist = 0;
    if (a.by_batch) {
        // persistent tiles (see gemm_bf3p_pipe_kernel): as many workgroups as the chip holds at once (LDS- and wave-limited per CU), a
        // multiple of 8 so that tile L keeps its XCD L % 8
        const int persist_env = 1;
        const int cus = bbdm_device_cus();
        const int by_lds = (int)((160 * 1024) / lds), by_waves = 16 / (WM * WN) > 0 ? 16 / (WM * WN) : 1;
        const int per_cu = by_lds < by_waves ? by_lds : by_waves;
        const unsigned resident = (unsigned)(cus * (per_cu > 0 ? per_cu : 1)) / 8 * 8;
        if (persist_env && resident >= 8 && grid.x > resident) {
            a.persist = 1;
            grid = dim3(resident);
        }
    }
    hipLaunchKernelGGL((gemm_bf3p_pipe_kernel<WM, WN, RES, NS, NP, R4>), grid, dim3(WM * WN * 64), lds, st, a);
    return BBDM_OK;
}

// Tile choice (env BBDM_BF3P_KERNEL / bbdm_debug_set_bf3p_kernel, the header's test-hook section): 6 (default) = 256 x 256 tiles where
// Cout fills 256-column tiles and the launch quantises well, else 256 x 128, 128 x 128 for the small layers; forced shapes for the
// A/B and for the tests (small test problems would otherwise only ever reach the 128 x 128 instantiation): 4 = 256 x 256 (256 x 128
// where Cout does not fill 256 columns), 5 = 256 x 128, 7 = 128 x 128.  Measured on the 42 Winograd layers of the C2 step,
// launch-weighted fp32-equivalent TFLOP/s (profiles/r03_bf3p_variants.txt; gemm_bf3.hip on the same GEMMs: 181): 5: 204 - 210,
// 4: 213 - 214, 6: 211 - 216.  (The non-pipelined two-stage kernel, 181 - 204, and a 3-stage LDS ring with counted vmcnt waits,
// 185 - 191, were measured in round 3 and deleted.)
#define g_bf3p_variant bbdm_option(BBDM_OPT_BF3P_KERNEL)

// ---- forward GEMMs: tile and split-K choice -------------------------------------------------------------------------------------
// Large problems (every Winograd layer of the 256^2 step) take 256 x 256 tiles, one 16-wave workgroup per CU.  SMALL problems -- the
// latent / 64^2-pixel configurations: a few hundred rows, K up to 2048 -- would leave most of the 256 CUs without a workgroup and
// stream their weights (read exactly once) through a handful of 2-deep prefetch queues: they take 128 x 128 tiles (4 waves) and, when
// even those do not give every CU a workgroup, split K: split z writes its partial sums to M[z][batch][T][ldo] and the consumer adds
// the partials in order (bbdm_winograd_output_splitk_stats_f32) -- deterministic.
namespace {
// (A cost model that chose shape AND split count per launch -- rounds on the busiest XCD x (chunks + 3) / shape efficiency + the HBM
// time of the extra partial sums -- was measured on every workload and deleted: it split the mid-size GEMMs of the latent models
// (e.g. 36 x [512 x 1024 x 1024]: 2 rounds of 256 x 256 tiles -> 3 half-length rounds) and lost everywhere, C3 16.95 -> 17.43 ms,
// C4 +2.4 ms: the last, partly filled round runs faster than the model assumes (fewer CUs share the power budget) and the split's
// extra partial sums cost the output transform more than the GEMM gains.  profiles/r03_bf3p_tile_choice.txt.)
int fwd_splits(int batch, long long rows, int CinPad, int Cout) {
    const int target = 256;
    const int nchunks = CinPad / KC;
    const long long base = (long long)batch * cdiv((int)rows, 128) * cdiv(Cout, 128);       // workgroups with the smallest tile
    if (target <= 0 || base * 2 > target) return 1;
    long long splits = (target + base - 1) / base;
    const long long max_splits = nchunks / 16 > 1 ? nchunks / 16 : 1;                  // >= 16 chunks (K = 256) per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits > 8) splits = 8;
    if (splits < 1) splits = 1;
    const int kps = (int)((nchunks + splits - 1) / splits);
    return (nchunks + kps - 1) / kps;
}

// rows: the rows actually computed (<= T, a multiple of 32; the tiles beyond them are neither launched nor stored); T: the row count
// the buffers are laid out for (per-batch strides).
// np = planes per operand: 3 (bf16x3) or 2 (the fp16 pair of h2_split.h; hA / hB = the bounds the operands were scaled by).
int bf3p_forward(const void* a_planes, const void* b_planes, const float* bias, const float* residual, int ldr, float* M, int ldo,
                 int batch, long long T, long long rows, int CinPad, int Cout, int splits, void* stream, int np = 3,
                 const float* hA = nullptr, const float* hB = nullptr, float gA = 1.f, float gB = 1.f) {
    BBDM_REQUIRE(a_planes && b_planes && M && batch > 0, "gemm_bf3p: null pointer / bad batch");
    BBDM_REQUIRE(np == 3 || (np == 2 && hA && hB), "gemm_h2p: the fp16-pair planes need the bounds of both operands");
    BBDM_REQUIRE(bbdm_gemm_bf3p_supported(T, CinPad, Cout), "gemm_bf3p: T=%lld CinPad=%d Cout=%d unsupported (T %% 256, CinPad %% 16)",
                 T, CinPad, Cout);
    BBDM_REQUIRE(rows > 0 && rows <= T && rows % 32 == 0, "gemm_bf3p: rows=%lld of T=%lld", rows, T);
    BBDM_REQUIRE((((uintptr_t)a_planes | (uintptr_t)b_planes) & 15) == 0 && ((uintptr_t)M & 3) == 0 && ldo >= Cout &&
                     (!residual || ldr >= Cout),
                 "gemm_bf3p: alignment / pitch");
    BBDM_REQUIRE(splits >= 1 && splits <= CinPad / KC && (splits == 1 || (!bias && !residual)),
                 "gemm_bf3p: splits=%d (partial sums carry neither bias nor residual)", splits);
    Bf3pArgs a;
    a.A = (const unsigned char*)a_planes; a.B = (const unsigned char*)b_planes; a.M = M;
    a.T = (int)rows; a.Cout = Cout; a.nchunks = CinPad / KC;
    a.rgs = bbdm_option(BBDM_OPT_BF3P_PAD_ROWS) ? (int)(T / 32) : (int)(rows / 32);      // (0: round 4's clamp to the last real group, A/B)
    const int CoutPad = cdiv(Cout, 128) * 128;
    a.tilesN = CoutPad / 128;
    a.az = (size_t)T * CinPad * 2 * np; a.bz = (size_t)CoutPad * CinPad * 2 * np; a.mz = (size_t)T * ldo; a.rz = (size_t)T * ldr;
    a.ldo = ldo; a.ldr = ldr; a.bias = bias; a.res = residual;
    a.hA = hA; a.hB = hB; a.gA = gA; a.gB = gB;
    a.kps = (a.nchunks + splits - 1) / splits;
    a.ksplits = (a.nchunks + a.kps - 1) / a.kps;
    BBDM_REQUIRE(a.ksplits == splits, "gemm_bf3p: %d splits of %d chunks leave an empty split", splits, a.nchunks);
    a.P = batch;
    const int nb = batch * splits;
    // (2 = any batch of 8 entries or more -- the 36 transform points of F(4x4,3x3) too: the kernels return early for the padding entries
    // of a batch % 8 != 0 -- measured on the LBBDM-f4 step: tile GEMMs 9.66 -> 9.39 ms; 1 = only multiples of 8, the round-2 rule)
    const int by_batch_env = 2;
    a.batch = nb;
    a.by_batch = (by_batch_env && nb >= 8 && (nb % 8 == 0 || by_batch_env == 2 || splits > 1)) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = CoutPad % 256 == 0;
    // workgroups a tile shape gives; below ~one per CU the next smaller shape takes over (pipe kernel only: it masks ragged row tiles)
    const int small_wg = 200;
    auto wgs = [&](int bm, int bn) { return (long long)nb * cdiv((int)rows, bm) * cdiv(CoutPad, bn); };
    int rc;
#define BBDM_BF3P_GO_NS(WM, WN, NS)                                                                                        \
    (np == 3 ? (residual ? bf3p_launch<WM, WN, true, NS, 3>(a, nb, st) : bf3p_launch<WM, WN, false, NS, 3>(a, nb, st))     \
             : (residual ? bf3p_launch<WM, WN, true, NS, 2>(a, nb, st) : bf3p_launch<WM, WN, false, NS, 2>(a, nb, st)))
#define BBDM_BF3P_GO(WM, WN) BBDM_BF3P_GO_NS(WM, WN, 2)
    if (g_bf3p_variant == 4) rc = wide ? BBDM_BF3P_GO(4, 4) : BBDM_BF3P_GO(4, 2);
    else if (g_bf3p_variant == 5) rc = BBDM_BF3P_GO(4, 2);
    else if (g_bf3p_variant == 7 || wgs(256, 128) < small_wg || rows <= 128) {        // (<= 128 rows: a 256-row tile would be half padding)
        // LDS stages of the 128 x 128 kernel: three when every workgroup of the launch gets a CU to itself -- nobody else's copies cover
        // the wait for its own, and two stages expose an HBM round trip per chunk (measured per layer, profiles/r04_small_gemm_ring.md:
        // 16 x [256 x 1024 x 1024] 61 -> 49 us, 16 x [512 x 512 x 512] 35 -> 30 us); with two or three workgroups per CU the third stage
        // loses 5 - 8 % (one workgroup's copies already overlap the other's MFMAs, and the ring's prologue waits for one chunk more),
        // and five stages never beat three.
        const int max_stages = 3;
        const long long w = (long long)cdiv((int)rows, 128) * cdiv(CoutPad, 128) * (a.by_batch ? (nb + 7) / 8 * 8 : nb);
        if (max_stages >= 3 && a.nchunks / splits >= 8 && w <= bbdm_device_cus()) rc = BBDM_BF3P_GO_NS(2, 2, 3);
        else rc = BBDM_BF3P_GO(2, 2);
    }
    else {
        // 256 x 256 (one workgroup per CU) vs 256 x 128 (two per CU): the CU that gets the most workgroups sets the time.  Mid-size
        // problems (the 16x16 / 32x32 levels of the latent models: 288 ... 1152 tiles of 256 x 256 on 256 CUs) lose up to half a
        // round to that quantisation; the smaller tile rounds in half steps at 0.94 of the large tile's rate (measured,
        // profiles/r03_bf3p_tile_choice.txt).  by_batch launches quantise per XCD (32 CUs, whole batch entries).
        auto rounds = [&](int bm, int bn) {
            const long long per_entry = (long long)cdiv((int)rows, bm) * cdiv(CoutPad, bn);
            return a.by_batch ? (double)((by_batch_slots(per_entry, nb) + 31) / 32) : (double)((per_entry * nb + 255) / 256);
        };
        const double t44 = wide ? rounds(256, 256) : 1e30, t42 = rounds(256, 128) * 0.5 / 0.94;
        rc = t44 <= t42 ? BBDM_BF3P_GO(4, 4) : BBDM_BF3P_GO(4, 2);
    }
#undef BBDM_BF3P_GO_NS
#undef BBDM_BF3P_GO
    if (rc != BBDM_OK) return rc;
    BBDM_CHECK_LAUNCH("gemm_bf3p");
    return BBDM_OK;
}
}  // namespace

// M[b][T][ldo] = A_b . B_b (+ bias) (+ residual[b][T][ldr]): A, B in the plane layout of the header
extern "C" int bbdm_gemm_bf3p_f32(const void* a_planes, const void* b_planes, const float* bias, const float* residual, int ldr,
                                  float* M, int ldo, int batch, long long T, int CinPad, int Cout, void* stream) {
    return bf3p_forward(a_planes, b_planes, bias, residual, ldr, M, ldo, batch, T, T, CinPad, Cout, 1, stream);
}

// Split-K forward GEMM for small problems: `rows` (<= T, a multiple of 32) rows are computed, split z < splits writes its partial sums
// to M[z][batch][T][ldo] (no bias / residual: the consumer adds the partials in order).  splits = bbdm_gemm_bf3p_fwd_splits(...) is the
// measured choice (1 for every problem that fills the chip without splitting); any 1 <= splits <= CinPad / 16 that leaves no split
// empty is accepted.
extern "C" int bbdm_gemm_bf3p_fwd_splits(int batch, long long rows, int CinPad, int Cout) {
    if (batch <= 0 || rows <= 0 || CinPad <= 0 || CinPad % KC || Cout <= 0) return 0;
    return fwd_splits(batch, rows, CinPad, Cout);
}
extern "C" int bbdm_gemm_bf3p_splitk_f32(const void* a_planes, const void* b_planes, float* M, int ldo, int batch, long long T,
                                         long long rows, int CinPad, int Cout, int splits, void* stream) {
    return bf3p_forward(a_planes, b_planes, nullptr, nullptr, 0, M, ldo, batch, T, rows, CinPad, Cout, splits, stream);
}

// ---- the fp16-pair ("h2") planes: two planes per operand under a power-of-two scale (h2_split.h; round 6) ----------------------------
// Layout as above with TWO units per chunk: A [batch][T / 32][nchunks][2][1 KB], B [batch][CoutPad / 32][nchunks][2][1 KB] -- 4 B per
// element.  Each operand comes with a BOUND (a device float >= max |x| over the whole operand): producers scale by 2^e,
// e = h2_exp_of_bound(bound), the GEMM's epilogue by 2^-(eA + eB).
namespace {
// max |x| into *bound (atomicMax on the bit pattern of a non-negative float: order-independent); *bound accumulates.  ONE atomic per
// workgroup and at most 512 workgroups: a device-scope atomic on one address retires at ~90 per microsecond (MI355X_MICROARCH.md,
// "dequeue"), and the first version -- one per wave of 4096 workgroups -- spent 190 us per launch on them (round 6: 8.2 ms of a C4 step).
__device__ __forceinline__ void h2_block_max_to(float m, float* bound) {
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(bound), __float_as_uint(m));
    }
}
__global__ void __launch_bounds__(256) h2_absmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ bound) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    h2_block_max_to(m, bound);
}
// ... of a [rows][C] view with pitch ld (a channel slice of a wider NHWC buffer); C % 4 == 0, 16-byte aligned rows.  A workgroup walks
// whole rows: thread t of a row's C / 4 float4 slots (no division in the loop)
__global__ void __launch_bounds__(256) h2_absmax_rows_kernel(const float* __restrict__ x, int ld, long long rows, int C4, int rpb,
                                                             float* __restrict__ bound) {
    float m = 0.f;
    const int c = threadIdx.x % C4, r0 = threadIdx.x / C4;          // rpb = 256 / C4 rows per pass (C4 <= 256), or 1 with a column loop
    if (rpb > 0) {
        if (r0 < rpb)
            for (long long r = (long long)blockIdx.x * rpb + r0; r < rows; r += (long long)gridDim.x * rpb) {
                const float4 v = *reinterpret_cast<const float4*>(x + r * ld + 4 * c);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
    } else {
        for (long long r = blockIdx.x; r < rows; r += gridDim.x)
            for (int cc = threadIdx.x; cc < C4; cc += 256) {
                const float4 v = *reinterpret_cast<const float4*>(x + r * ld + 4 * cc);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
    }
    h2_block_max_to(m, bound);
}
// bf3p_pack_b_kernel for the fp16 pair
__global__ void h2p_pack_b_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, const float* __restrict__ bound,
                                  float gain, size_t batch_chunks, int nchunks, int CoutPad) {
    const float sc = h2_pow2(h2_exp_of_bound(*bound * gain));
    const size_t pairs = batch_chunks * CoutPad * (KC / 2);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        const int kp = (int)(i % (KC / 2));
        size_t t = i / (KC / 2);
        const int co = (int)(t % CoutPad);
        const size_t bc = t / CoutPad;
        const size_t b = bc / nchunks;
        const int chunk = (int)(bc % nchunks);
        const float2 v = *reinterpret_cast<const float2*>(src + (bc * CoutPad + co) * KC + kp * 2);
        unsigned p1, p2;
        h2_split2(v.x * sc, v.y * sc, p1, p2);
        unsigned char* d = dst + (((b * (CoutPad / 32) + co / 32) * nchunks + chunk) * 2) * (size_t)UNIT + unit_off(co & 31, kp * 2);
        *reinterpret_cast<unsigned*>(d) = p1;
        *reinterpret_cast<unsigned*>(d + UNIT) = p2;
    }
}
// bf3p_split_rows_kernel for the fp16 pair
__global__ void __launch_bounds__(256) h2p_split_rows_kernel(const float* __restrict__ x, int ld, size_t xz, unsigned char* __restrict__ dst,
                                                              size_t dz, const float* __restrict__ bound, int T, int nchunks) {
    const float sc = h2_pow2(h2_exp_of_bound(*bound));
    const int chunk = (int)blockIdx.x % nchunks, g = (int)blockIdx.x / nchunks, b = (int)blockIdx.y;
    const int r = threadIdx.x >> 3, kp = threadIdx.x & 7;
    const int row = g * 32 + r;
    float2 v = make_float2(0.f, 0.f);
    if (row < T) v = *reinterpret_cast<const float2*>(x + (size_t)b * xz + (size_t)row * ld + chunk * KC + kp * 2);
    unsigned p1, p2;
    h2_split2(v.x * sc, v.y * sc, p1, p2);
    unsigned char* d = dst + (size_t)b * dz + (((size_t)g * nchunks + chunk) * 2) * UNIT + unit_off(r, kp * 2);
    *reinterpret_cast<unsigned*>(d) = p1;
    *reinterpret_cast<unsigned*>(d + UNIT) = p2;
}
}  // namespace

extern "C" size_t bbdm_gemm_h2p_a_bytes(int batch, long long T, int CinPad) {
    return (size_t)batch * (size_t)((T + 255) / 256 * 256) * (size_t)CinPad * 4;
}
extern "C" size_t bbdm_gemm_h2p_b_bytes(int batch, int CinPad, int Cout) {
    return (size_t)batch * (size_t)(cdiv(Cout, 128) * 128) * (size_t)CinPad * 4;
}
extern "C" int bbdm_absmax_f32(const float* x, long long n, float* bound, void* stream) {
    BBDM_REQUIRE(x && bound && n > 0, "absmax: bad args");
    size_t blocks = ((size_t)n + 4095) / 4096;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(h2_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, bound);
    BBDM_CHECK_LAUNCH("absmax");
    return BBDM_OK;
}
extern "C" int bbdm_absmax_rows_f32(const float* x, int ldx, long long rows, int C, float* bound, void* stream) {
    BBDM_REQUIRE(x && bound && rows > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "absmax_rows: bad args");
    const int C4 = C / 4, rpb = C4 <= 256 ? 256 / C4 : 0;
    long long blocks = rpb ? (rows + 4 * rpb - 1) / (4 * rpb) : rows;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(h2_absmax_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C4, rpb, bound);
    BBDM_CHECK_LAUNCH("absmax_rows");
    return BBDM_OK;
}
extern "C" int bbdm_gemm_h2p_pack_b_f32(const float* packed_f32, void* b_planes, const float* bound, float gain, int batch, int CinPad,
                                        int Cout, void* stream) {
    BBDM_REQUIRE(packed_f32 && b_planes && bound && batch > 0 && CinPad > 0 && CinPad % KC == 0 && Cout > 0, "gemm_h2p_pack_b: bad args");
    const int CoutPad = cdiv(Cout, 128) * 128, nchunks = CinPad / KC;
    const size_t pairs = (size_t)batch * nchunks * CoutPad * (KC / 2);
    size_t blocks = (pairs + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(h2p_pack_b_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed_f32,
                       (unsigned char*)b_planes, bound, gain, (size_t)batch * nchunks, nchunks, CoutPad);
    BBDM_CHECK_LAUNCH("gemm_h2p_pack_b");
    return BBDM_OK;
}
extern "C" int bbdm_gemm_h2p_split_rows_f32(const float* x, int ldx, void* a_planes, const float* bound, int batch, long long T,
                                            int CinPad, void* stream) {
    BBDM_REQUIRE(x && a_planes && bound && batch > 0 && T > 0 && CinPad > 0 && CinPad % KC == 0 && ldx >= CinPad && ldx % 2 == 0,
                 "gemm_h2p_split_rows: bad args");
    BBDM_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)a_planes & 15) == 0, "gemm_h2p_split_rows: alignment");
    const long long Tp = (T + 255) / 256 * 256;
    const int nchunks = CinPad / KC;
    BBDM_REQUIRE((Tp / 32) * nchunks < (1ll << 31) && batch < 65536, "gemm_h2p_split_rows: too large");
    hipLaunchKernelGGL(h2p_split_rows_kernel, dim3((unsigned)((Tp / 32) * nchunks), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, (size_t)T * ldx, (unsigned char*)a_planes, (size_t)Tp * CinPad * 4, bound, (int)T,
                       nchunks);
    BBDM_CHECK_LAUNCH("gemm_h2p_split_rows");
    return BBDM_OK;
}
// M[b][T][ldo] = A_b . B_b (+ bias) (+ residual) on the fp16-pair planes; bound_a / bound_b: the device floats the producers scaled by
extern "C" int bbdm_gemm_h2p_f32(const void* a_planes, const void* b_planes, const float* bound_a, const float* bound_b,
                                 const float* bias, const float* residual, int ldr, float* M, int ldo, int batch, long long T,
                                 int CinPad, int Cout, void* stream) {
    return bf3p_forward(a_planes, b_planes, bias, residual, ldr, M, ldo, batch, T, T, CinPad, Cout, 1, stream, 2, bound_a, bound_b);
}
extern "C" int bbdm_gemm_h2p_splitk_f32(const void* a_planes, const void* b_planes, const float* bound_a, const float* bound_b, float* M,
                                        int ldo, int batch, long long T, long long rows, int CinPad, int Cout, int splits, void* stream) {
    return bf3p_forward(a_planes, b_planes, nullptr, nullptr, 0, M, ldo, batch, T, rows, CinPad, Cout, splits, stream, 2, bound_a,
                        bound_b);
}
// (winograd.hip: the A planes were scaled by the bound of the transform's INPUT times its gain)
int bbdm_gemm_h2p_gain_splitk(const void* a_planes, const void* b_planes, const float* bound_a, float gain_a, const float* bound_b,
                              float gain_b, float* M, int ldo, int batch, long long T, long long rows, int CinPad, int Cout, int splits,
                              void* stream) {
    return bf3p_forward(a_planes, b_planes, nullptr, nullptr, 0, M, ldo, batch, T, rows, CinPad, Cout, splits, stream, 2, bound_a,
                        bound_b, gain_a, gain_b);
}

// ---- C = A^T B with the contraction over the ROWS of both operands (the Winograd-domain weight gradient dU_xi = V_xi^T dM_xi, tiles
// contracted; replaces gemm_tn_f32 of winograd_wgrad.hip where its operands come as planes) ------------------------------------------
// at_planes: [batch][M / 32][K / 16][3][1 KB] -- units whose rows are the M index and whose k is the contraction index, i.e. the
// TRANSPOSED operands as bbdm_winograd_input_bf3p_tr_f32 / bbdm_winograd_dy_transform_bf3p_f32 write them; bt_planes likewise
// [batch][NPad128 / 32][K / 16][3][1 KB].  It is the kernel above verbatim (rows = M, columns = N, chunks = K / 16) plus split-K:
// C[z][b][M][N], z < bbdm_gemm_bf3p_tn_splits, partial sums over disjoint K ranges (added in a fixed order by the consumer:
// bbdm_winograd_wgrad_finish_f32 -- deterministic).  K % 16 == 0 (zero-padded), M % 32 == 0, N % 4 == 0.
namespace {
struct TnSplit { int splits, kps, wn; };
// split count for one tile width (256- or 128-column tiles, 256 rows): two workgroups' worth of tiles per CU when K allows
TnSplit tn_split_for(int batch, long long K, int M, int N, int bn) {
    const int nchunks = (int)(K / KC);
    const int NPad = cdiv(N, 128) * 128;
    const long long base = (long long)batch * cdiv(M, 256) * cdiv(NPad, bn);
    long long splits = (512 + base - 1) / base;
    const long long max_splits = nchunks / 16 > 1 ? nchunks / 16 : 1;     // >= 16 chunks per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    TnSplit t;
    t.kps = (int)((nchunks + splits - 1) / splits);
    t.splits = (nchunks + t.kps - 1) / t.kps;
    t.wn = bn / 64;
    return t;
}
// ... and the tile width: the CU (of the XCD: by_batch launches deal whole batch entries to XCDs) with the most workgroups sets the
// time = its workgroup count x chunks per workgroup x the tile's cost (256 x 128: half, at 0.94 of the large tile's rate) -- the
// forward GEMM's rule (bf3p_forward), here with the split count in the model.  Deterministic in (batch, K, M, N): the consumer asks
// bbdm_gemm_bf3p_tn_splits for the same answer.
TnSplit tn_split(int batch, long long K, int M, int N) {
    const int NPad = cdiv(N, 128) * 128;
    const TnSplit narrow = tn_split_for(batch, K, M, N, 128);
    if (NPad % 256) return narrow;
    const TnSplit wide = tn_split_for(batch, K, M, N, 256);
    auto cost = [&](const TnSplit& t, int bn) {
        const long long per_entry = (long long)cdiv(M, 256) * cdiv(NPad, bn), nb = (long long)batch * t.splits;
        const double rounds = nb >= 8 ? (double)((per_entry * ((nb + 7) / 8) + 31) / 32) : (double)((per_entry * nb + 255) / 256);
        return rounds * (t.kps + 4) * (bn == 256 ? 1.0 : 0.5 / 0.94);           // (+ 4 chunks: prologue / epilogue of a workgroup)
    };
    return cost(wide, 256) <= cost(narrow, 128) ? wide : narrow;
}
}  // namespace

extern "C" size_t bbdm_gemm_bf3p_tn_at_bytes(int batch, long long K, int M) {
    return (size_t)batch * (size_t)((M + 31) / 32 * 32) * (size_t)((K + 255) / 256 * 256) * 6;
}
extern "C" size_t bbdm_gemm_bf3p_tn_bt_bytes(int batch, long long K, int N) {
    return (size_t)batch * (size_t)(cdiv(N, 128) * 128) * (size_t)((K + 255) / 256 * 256) * 6;
}
extern "C" int bbdm_gemm_bf3p_tn_supported(long long K, int M, int N) {
    return K > 0 && K % 256 == 0 && K < (1ll << 31) && M > 0 && M % 32 == 0 && N > 0 && N % 4 == 0;
}
extern "C" int bbdm_gemm_bf3p_tn_splits(int batch, long long K, int M, int N) {
    if (batch <= 0 || !bbdm_gemm_bf3p_tn_supported(K, M, N)) return 0;
    return tn_split(batch, K, M, N).splits;
}

namespace {
int bf3p_tn(const void* at_planes, const void* bt_planes, float* C, int batch, long long K, int M, int N, void* stream, int np,
            const float* hA, float gA, const float* hB, float gB) {
    BBDM_REQUIRE(at_planes && bt_planes && C && batch > 0, "gemm_bf3p_tn: null pointer / bad batch");
    BBDM_REQUIRE(bbdm_gemm_bf3p_tn_supported(K, M, N), "gemm_bf3p_tn: K=%lld M=%d N=%d unsupported (K %% 256, M %% 32, N %% 4)", K, M, N);
    BBDM_REQUIRE((((uintptr_t)at_planes | (uintptr_t)bt_planes) & 15) == 0 && ((uintptr_t)C & 3) == 0, "gemm_bf3p_tn: alignment");
    BBDM_REQUIRE(np == 3 || (np == 2 && hA && hB), "gemm_h2p_tn: the fp16-pair planes need the bounds of both operands");
    const TnSplit sp = tn_split(batch, K, M, N);
    Bf3pArgs a;
    a.A = (const unsigned char*)at_planes; a.B = (const unsigned char*)bt_planes; a.M = C;
    a.T = M; a.Cout = N; a.nchunks = (int)(K / KC);
    a.rgs = (M + 31) / 32;
    const int NPad = cdiv(N, 128) * 128;
    a.tilesN = NPad / 128;
    a.az = (size_t)((M + 31) / 32 * 32) * K * 2 * np; a.bz = (size_t)NPad * K * 2 * np; a.mz = (size_t)M * N; a.rz = 0;
    a.ldo = N; a.ldr = 0; a.bias = nullptr; a.res = nullptr;
    a.ksplits = sp.splits; a.kps = sp.kps; a.P = batch;
    a.hA = hA; a.hB = hB; a.gA = gA; a.gB = gB;
    const int nb = batch * sp.splits;
    const int by_batch_env = 1;
    a.batch = nb;
    a.by_batch = (by_batch_env && nb >= 8) ? 1 : 0;              // (the pipe kernel returns early for the padding entries of a batch % 8 != 0)
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (np == 3) rc = sp.wn == 4 ? bf3p_launch<4, 4, false>(a, nb, st) : bf3p_launch<4, 2, false>(a, nb, st);
    else rc = sp.wn == 4 ? bf3p_launch<4, 4, false, 2, 2>(a, nb, st) : bf3p_launch<4, 2, false, 2, 2>(a, nb, st);
    if (rc != BBDM_OK) return rc;
    BBDM_CHECK_LAUNCH("gemm_bf3p_tn");
    return BBDM_OK;
}
}  // namespace

extern "C" int bbdm_gemm_bf3p_tn_f32(const void* at_planes, const void* bt_planes, float* C, int batch, long long K, int M, int N,
                                     void* stream) {
    return bf3p_tn(at_planes, bt_planes, C, batch, K, M, N, stream, 3, nullptr, 1.f, nullptr, 1.f);
}
// ... on the fp16-pair planes (round 6; 4 B per element: bbdm_gemm_h2p_tn_at_bytes / _bt_bytes; same split rule): the operands were
// scaled by bound_a x gain_a and bound_b x gain_b (the Winograd-domain weight gradient: the GroupNorm bound of the layer's input x
// bbdm_winograd_input_gain(m), the measured maximum of dY x bbdm_winograd_dy_gain(m))
extern "C" size_t bbdm_gemm_h2p_tn_at_bytes(int batch, long long K, int M) {
    return (size_t)batch * (size_t)((M + 31) / 32 * 32) * (size_t)((K + 255) / 256 * 256) * 4;
}
extern "C" size_t bbdm_gemm_h2p_tn_bt_bytes(int batch, long long K, int N) {
    return (size_t)batch * (size_t)(cdiv(N, 128) * 128) * (size_t)((K + 255) / 256 * 256) * 4;
}
extern "C" int bbdm_gemm_h2p_tn_f32(const void* at_planes, const void* bt_planes, float* C, int batch, long long K, int M, int N,
                                    const float* bound_a, float gain_a, const float* bound_b, float gain_b, void* stream) {
    return bf3p_tn(at_planes, bt_planes, C, batch, K, M, N, stream, 2, bound_a, gain_a, bound_b, gain_b);
}

// ---- fp32 A operand on the pipelined kernel (gemm_bf3q_pipe_kernel) ---------------------------------------------------------------------
namespace {
template <int WM, int WN, bool RES, int NP = 3>
int bf3q_launch(Bf3pArgs& a, const float* Af, int lda, int batch, hipStream_t st) {
    static bool attr_set_dev[BBDM_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[bbdm_device_slot()];
    const size_t lds = 2 * (size_t)(WM * 2 * NP + WN * 2 * NP) * UNIT;
    const void* fn = reinterpret_cast<const void*>(gemm_bf3q_pipe_kernel<WM, WN, RES, NP>);
    if (!attr_set) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            bbdm_set_error("gemm_bf3q: hipFuncSetAttribute(%zu B LDS) failed", lds);
            return BBDM_E_LAUNCH;
        }
        attr_set = true;
    }
    const long long blocks = (((long long)a.T + WM * 64 - 1) / (WM * 64)) * (a.tilesN * 2 / WN);
    BBDM_REQUIRE(blocks * ((batch + 7) / 8) * 8 < (1ll << 31), "gemm_bf3q: too many tiles");
    a.tiles = (int)blocks;
    a.persist = 0;
    const dim3 grid = a.by_batch ? dim3((unsigned)(8 * blocks * ((batch + 7) / 8))) : dim3((unsigned)blocks, 1, batch);
    hipLaunchKernelGGL((gemm_bf3q_pipe_kernel<WM, WN, RES, NP>), grid, dim3(WM * WN * 64), lds, st, a, Af, lda);
    return BBDM_OK;
}
}  // namespace

// out[pixels][ldo] = x[pixels][ldx] . W^T + bias (+ residual): the 1x1 convolutions / Linears of bbdm_conv1x1_bf3_f32 (same call
// sites, same arithmetic bit for bit) on the pipelined kernel.  b_planes = bbdm_gemm_bf3p_pack_b_f32(batch = 1) of the buffer
// bbdm_conv_pack_weight_f32(ks = 1) filled.  Any pixel count (a ragged last row tile re-reads the last row and masks its stores), CinPad a
// multiple of 16.
extern "C" int bbdm_conv1x1_bf3q_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr,
                                     float* out, int ldo, long long pixels, int CinPad, int Cout, void* stream) {
    BBDM_REQUIRE(x && b_planes && out, "conv1x1_bf3q: null pointer");
    BBDM_REQUIRE(pixels > 0 && pixels < (1ll << 31) && CinPad > 0 && CinPad % KC == 0 && Cout > 0 && Cout % 4 == 0,
                 "conv1x1_bf3q: pixels=%lld CinPad=%d Cout=%d unsupported", pixels, CinPad, Cout);
    BBDM_REQUIRE(ldx % 4 == 0 && ldx >= CinPad && ldo >= Cout && (!residual || ldr >= Cout) &&
                     (((uintptr_t)x | (uintptr_t)b_planes) & 15) == 0,
                 "conv1x1_bf3q: bad pitch / alignment");
    Bf3pArgs a;
    a.A = nullptr; a.B = (const unsigned char*)b_planes; a.M = out;
    a.T = (int)pixels; a.Cout = Cout; a.nchunks = CinPad / KC;
    a.rgs = 0;
    const int CoutPad = cdiv(Cout, 128) * 128;
    a.tilesN = CoutPad / 128;
    a.az = 0; a.bz = 0; a.mz = 0; a.rz = 0;
    a.ldo = ldo; a.ldr = ldr; a.bias = bias; a.res = residual;
    a.ksplits = 1; a.kps = a.nchunks; a.P = 1; a.batch = 1; a.by_batch = 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (CoutPad % 256 == 0) rc = residual ? bf3q_launch<3, 4, true>(a, x, ldx, 1, st) : bf3q_launch<3, 4, false>(a, x, ldx, 1, st);
    else rc = residual ? bf3q_launch<3, 2, true>(a, x, ldx, 1, st) : bf3q_launch<3, 2, false>(a, x, ldx, 1, st);
    if (rc != BBDM_OK) return rc;
    BBDM_CHECK_LAUNCH("conv1x1_bf3q");
    return BBDM_OK;
}

// The same product on the fp16-pair planes (round 6; h2_split.h): b_planes = bbdm_gemm_h2p_pack_b_f32(batch = 1) under wbound (the exact
// maximum of the packed weights), xbound: a device float >= max |x| over the pixels and channels read.  256-row tiles.
extern "C" int bbdm_conv1x1_h2q_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr,
                                    float* out, int ldo, long long pixels, int CinPad, int Cout, const float* xbound, const float* wbound,
                                    void* stream) {
    BBDM_REQUIRE(x && b_planes && out && xbound && wbound, "conv1x1_h2q: null pointer");
    BBDM_REQUIRE(pixels > 0 && pixels < (1ll << 31) && CinPad > 0 && CinPad % KC == 0 && Cout > 0 && Cout % 4 == 0,
                 "conv1x1_h2q: pixels=%lld CinPad=%d Cout=%d unsupported", pixels, CinPad, Cout);
    BBDM_REQUIRE(ldx % 4 == 0 && ldx >= CinPad && ldo >= Cout && (!residual || ldr >= Cout) &&
                     (((uintptr_t)x | (uintptr_t)b_planes) & 15) == 0,
                 "conv1x1_h2q: bad pitch / alignment");
    Bf3pArgs a;
    a.A = nullptr; a.B = (const unsigned char*)b_planes; a.M = out;
    a.T = (int)pixels; a.Cout = Cout; a.nchunks = CinPad / KC;
    a.rgs = 0;
    const int CoutPad = cdiv(Cout, 128) * 128;
    a.tilesN = CoutPad / 128;
    a.az = 0; a.bz = 0; a.mz = 0; a.rz = 0;
    a.ldo = ldo; a.ldr = ldr; a.bias = bias; a.res = residual;
    a.ksplits = 1; a.kps = a.nchunks; a.P = 1; a.batch = 1; a.by_batch = 0;
    a.hA = xbound; a.hB = wbound; a.gA = 1.f;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (CoutPad % 256 == 0) rc = residual ? bf3q_launch<4, 4, true, 2>(a, x, ldx, 1, st) : bf3q_launch<4, 4, false, 2>(a, x, ldx, 1, st);
    else rc = residual ? bf3q_launch<4, 2, true, 2>(a, x, ldx, 1, st) : bf3q_launch<4, 2, false, 2>(a, x, ldx, 1, st);
    if (rc != BBDM_OK) return rc;
    BBDM_CHECK_LAUNCH("conv1x1_h2q");
    return BBDM_OK;
}

// The same product for SMALL problems (gemm_bf3s_kernel above: 64 x 64 tiles, 64 channels per iteration, one launch, no split-K):
// same arguments, same b_planes, same bits as bbdm_conv1x1_bf3q_f32.  The caller chooses (unet.py: below BBDM_BF3_MIN_TILES tiles of
// 256 x 128); any pixel count, CinPad a multiple of 64.
namespace {
template <int NP>
int conv1x1_small(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr, float* out, int ldo,
                  long long pixels, int CinPad, int Cout, const float* xbound, const float* wbound, void* stream) {
    BBDM_REQUIRE(x && b_planes && out, "conv1x1_bf3s: null pointer");
    BBDM_REQUIRE(pixels > 0 && pixels < (1ll << 31) && CinPad > 0 && CinPad % 64 == 0 && Cout > 0 && Cout % 4 == 0,
                 "conv1x1_bf3s: pixels=%lld CinPad=%d Cout=%d unsupported (CinPad %% 64)", pixels, CinPad, Cout);
    BBDM_REQUIRE(ldx % 4 == 0 && ldx >= CinPad && ldo >= Cout && (!residual || ldr >= Cout) &&
                     (((uintptr_t)x | (uintptr_t)b_planes) & 15) == 0,
                 "conv1x1_bf3s: bad pitch / alignment");
    Bf3pArgs a;
    a.A = nullptr; a.B = (const unsigned char*)b_planes; a.M = out;
    a.T = (int)pixels; a.Cout = Cout; a.nchunks = CinPad / KC;
    a.rgs = 0;
    a.tilesN = cdiv(Cout, 128);
    a.az = 0; a.bz = 0; a.mz = 0; a.rz = 0;
    a.ldo = ldo; a.ldr = ldr; a.bias = bias; a.res = residual;
    a.ksplits = 1; a.kps = a.nchunks; a.P = 1; a.batch = 1; a.persist = 0; a.tiles = 0;
    a.hA = xbound; a.hB = wbound;
    const int tilesM = cdiv((int)pixels, 64), tilesN = cdiv(Cout, 64);
    a.by_batch = tilesN % 8 == 0 ? 1 : 0;
    constexpr int NS = 3;      // (four stages = all 160 KB of LDS, measured: no faster -- the chain is bound by what one CU's LDS-DMA path moves,
                               // ~40 GB/s, not by the distance of the prefetch: profiles/r04_small_1x1.md)
    const size_t lds = NS * (size_t)(64 * 64 * 4 + 2 * 4 * NP * UNIT);
    static bool attr_set_dev[BBDM_MAX_DEVICES][2] = {};
    bool& attr_set = attr_set_dev[bbdm_device_slot()][residual ? 1 : 0];
    if (!attr_set) {
        const void* fn = residual ? reinterpret_cast<const void*>(gemm_bf3s_kernel<true, NS, NP>) : reinterpret_cast<const void*>(gemm_bf3s_kernel<false, NS, NP>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            bbdm_set_error("conv1x1_bf3s: hipFuncSetAttribute(%zu B LDS) failed", lds);
            return BBDM_E_LAUNCH;
        }
        attr_set = true;
    }
    BBDM_REQUIRE((long long)tilesM * tilesN < (1ll << 31), "conv1x1_bf3s: too many tiles");
    const dim3 grid((unsigned)(tilesM * tilesN));
    hipStream_t st = (hipStream_t)stream;
    if (residual) hipLaunchKernelGGL((gemm_bf3s_kernel<true, NS, NP>), grid, dim3(256), lds, st, a, x, ldx);
    else hipLaunchKernelGGL((gemm_bf3s_kernel<false, NS, NP>), grid, dim3(256), lds, st, a, x, ldx);
    BBDM_CHECK_LAUNCH("conv1x1_bf3s");
    return BBDM_OK;
}
}  // namespace

extern "C" int bbdm_conv1x1_bf3s_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr,
                                     float* out, int ldo, long long pixels, int CinPad, int Cout, void* stream) {
    return conv1x1_small<3>(x, ldx, b_planes, bias, residual, ldr, out, ldo, pixels, CinPad, Cout, nullptr, nullptr, stream);
}
// ... on the fp16-pair planes (round 6): the arguments and bounds of bbdm_conv1x1_h2q_f32, the tile shape and size range of
// bbdm_conv1x1_bf3s_f32 (CinPad a multiple of 64); a third fewer weight bytes for launches that are bound by the weight stream
extern "C" int bbdm_conv1x1_h2s_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr,
                                    float* out, int ldo, long long pixels, int CinPad, int Cout, const float* xbound, const float* wbound,
                                    void* stream) {
    BBDM_REQUIRE(xbound && wbound, "conv1x1_h2s: null bound");
    return conv1x1_small<2>(x, ldx, b_planes, bias, residual, ldr, out, ldo, pixels, CinPad, Cout, xbound, wbound, stream);
}
