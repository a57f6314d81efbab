// attention.hip -- fp32 QKV self-attention with a streaming (online) softmax on the f32-input matrix core.
//
// Replaces QKVAttentionLegacy.forward (openaimodel.py:359-375) and QKVAttention.forward (:398-413):
//     w = softmax_over_s( (q*s)^T (k*s) ),  a = v w^T,   s = ch^-1/4
// without ever materialising the [N*heads, T, T] score tensor (17 GB at 256^2 / batch 16).
//
// One workgroup = NW waves (8: 256 consecutive queries of one (image, head); 4 for short sequences); each wave owns 32 queries.
// Workgroup ids are dealt so that every query block of one (image, head) runs on ONE XCD: its K / V (2 MB at T = 4096) cross the
// fabric into that L2 once instead of once per XCD (measured before: 4.8 GB of fabric traffic per launch against 1.07 GB).
// Everything is computed TRANSPOSED so that a query is a lane (column) in every MFMA result:
//   S^T[key, query] = K[key, :] . Q^T[:, query]     A = K tile from LDS, B = Q^T held in 2*CH/4 VGPRs
//   O^T[c,   query] += V^T[c, key] . P^T[key, query] A = V^T read from LDS, B = P^T = exp(S^T - m) IN PLACE:
// the 32x32 C/D layout gives lane l the keys {(r&3)+8(r>>2)+4(l>>5)} of query l&31 in register r, and the 32x32x2
// B operand wants, for its two k, lanes <32 -> k0 and lanes >=32 -> k1.  Choosing (k0,k1) = (key_lo(r), key_lo(r)+4)
// for step r makes register r of P^T exactly the B operand of step r -- no LDS round trip, no shuffles; row
// max / sum are 15 VALU ops + one cross-half exchange.  The running max / sum / rescale are per-lane scalars.
#include "bf3_split.h"
#include "h2_split.h"
#include "lds_dma.h"
#include <stdlib.h>

namespace {

constexpr int KT = 32;        // keys per tile
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;

// BQ: S^T = K Q^T on the BF16 matrix core with fp32 accuracy (bf3_split.h: both operands split exactly into three bf16 planes,
// six product terms, fp32 accumulate): 6 x CH/16 MFMAs of 32 cycles per key tile instead of CH/2 of 64 -- 2.7x less matrix-core
// time for half of the attention FLOPs.  K is split while it is staged (planes [3][key][CH] bf16 in LDS, +16 B row pad: the 8 rows
// of a ds_read_b128 group fall on distinct 16-B slots), Q once per workgroup; the bf16 MFMA's C/D layout is the f32 one, so the
// softmax and the P V product below (f32 MFMA: its B operand is the P registers in place) are untouched.
// BV: O^T += V^T P^T on the BF16 matrix core with fp32 accuracy as well: V is split into three bf16 planes while it is staged
// (row-major [plane][16-channel subtile][key][16 ch]: 8-byte writes, like K) and reaches the MFMA's A operand -- rows = channels,
// k = keys -- through ds_read_b64_tr_b16, the LDS transpose read: a 16-lane group reads a [4 keys][16 channels] block and lane i
// receives the 4 keys of channel i.  P^T = exp(S^T - m) is the B operand IN PLACE as before: register r of lane (query, hi) holds
// key (r & 3) + 8 (r >> 2) + 4 hi, so registers 8 ks .. 8 ks + 7 -- keys {0-3, 8-11} + 4 hi + 16 ks -- are the lane's 8 k-slots of
// K-step ks once split and packed pairwise (lane-local: no shuffles), and the transpose reads fetch exactly those keys.
// 6 x 2 x CH/32 MFMAs of 32 cycles per key tile instead of 16 x CH/32 of 64.
// PIPE (round 5; BQ && BV): the VALU work that does not depend on the tile's own scores is issued BETWEEN the MFMAs instead of in
// phases of its own.  The phased loop compiled to [24 MFMAs] [~160 VALU: softmax, P split] [12 MFMAs] [34 VALU] [12 MFMAs] [~80 VALU + 12
// ds_write: K / V split of the next tile], and the kernel's time was the SUM of its matrix-pipe time and its VALU issue time (a wave's
// VALU instructions hide under its own or another wave's MFMAs only ~5 per 32-cycle MFMA, and only when they are there to be issued).
// Now the split + LDS stores of tile t + 1 are dealt between the 24 Q K^T MFMAs of tile t (one stage of 3 - 8 instructions behind each
// MFMA, fenced with sched_barrier), K / V of tile t + 2 are requested right behind them into the same registers (they land under the
// softmax and the P V product), and the P split of the second key half is dealt between the first MFMAs of the first half's P V.  Same arithmetic, same bits (tests: attention with option "attn_pipe" 0 / 1).
template <int CH, bool BQ, bool BV, int NW, bool PIPE = false>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 3 : 2) attn_fwd_kernel(const float* __restrict__ qsrc, int ldq, int hsq,
                                                       const float* __restrict__ ksrc, const float* __restrict__ vsrc, int ldkv,
                                                       int hskv, float* __restrict__ out, int ldo, float* __restrict__ lse,
                                                       int Tq, int T, int heads, int nheads_total, float qscale, float scale) {
    // q: [N][Tq][ldq], head h at channel h * hsq;  k, v: [N][T][ldkv], head h at channel h * hskv (T = number of keys).
    // qscale / scale multiply q / k while they are loaded (legacy: ch^-1/4 each; CrossAttention: ch^-1/2 and 1).
    constexpr int KPITCH = CH + 4;                 // K tile pitch: b128 reads by 32 keys conflict-free
    constexpr int VPITCH = CH < 32 ? 32 : CH;      // V tile pitch (lanes sweep channels)
    constexpr int CT = (CH + 31) / 32;             // 32-row channel tiles of O^T
    constexpr int KG = CH / 8;                     // k-groups of 8 channels for QK^T
    constexpr int KV4 = KT * CH / 4;               // float4 per K (or V) tile
    constexpr int NTHR = NW * 64, QB = NW * 32;
    constexpr int SLOTS = (KV4 + NTHR - 1) / NTHR;
    static_assert(!BV || (BQ && CH % 32 == 0), "the bf16x3 P V path needs whole 32-channel tiles and the split K path");
    static_assert(!PIPE || (BQ && BV), "the interleaved loop exists for the bf16x3 path");

    constexpr int KS = CH / 16;                    // BQ: MFMA K-steps of 16 channels
    constexpr int KROWB = CH * 2 + 16;             // BQ: bytes per key row of one bf16 plane
    constexpr int KPLANE = KT * KROWB;             // BQ: bytes per plane
    constexpr int KSTAGE = BQ ? 3 * KPLANE / 4 : KT * KPITCH;      // floats per K stage

    constexpr int VSUB = KT * 32 + 128;            // BV: bytes from the first [32 keys][16 ch] bf16 subtile of a 32-channel tile to the
                                                   //     second (+128: the two subtiles a 32-lane half reads fall on different bank halves)
    constexpr int VCT = 2 * KT * 32 + 128;         // BV: bytes per 32-channel tile (two subtiles)
    constexpr int VPLANE = (CH / 32) * VCT;        // BV: bytes per plane
    constexpr int VSTAGE = BV ? 3 * VPLANE / 4 : KT * VPITCH;      // floats per V stage
    constexpr int OPITCH = CH + 1;                 // epilogue: O^T / l staged [query][CH + 1] through the same LDS
    constexpr int SMEM = 2 * KSTAGE + 2 * VSTAGE > QB * OPITCH ? 2 * KSTAGE + 2 * VSTAGE : QB * OPITCH;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* kbuf = smem;
    float* vbuf = smem + 2 * KSTAGE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    const int qblocks = (Tq + QB - 1) / QB;
    qscale *= LOG2E;                               // softmax in base 2: exp(x) = 2^(x log2 e)
    // XCD x (block id % 8) owns the (image, head) pairs x, x + 8, ...: all query blocks of a pair share its K / V through one L2
    const int L = (int)blockIdx.x, slot = L >> 3;
    const int qb = slot % qblocks;
    const int nh = (L & 7) + 8 * (slot / qblocks);
    if (nh >= nheads_total) return;
    const int h = nh % heads, n = nh / heads;
    const float* qbase = qsrc + (size_t)n * Tq * ldq + h * hsq;
    const float* kbase = ksrc + (size_t)n * T * ldkv + h * hskv;
    const float* vbase = vsrc + (size_t)n * T * ldkv + h * hskv;

    // ---- Q^T fragment: lane holds q[c] for c = kg*8 + hi*4 + j -------------------------------------------------
    const int q = qb * QB + wave * 32 + lq;
    float4 qf[BQ ? 1 : KG];
    bf16x8 qb3[BQ ? KS : 1][3];                    // BQ: lane holds q[c], c = ks*16 + hi*8 + 0..7, as three bf16 planes
    if (BQ) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (q < Tq) {
                v0 = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + ks * 16 + hi * 8);
                v1 = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + ks * 16 + hi * 8 + 4);
            }
            v0 = make_float4(v0.x * qscale, v0.y * qscale, v0.z * qscale, v0.w * qscale);
            v1 = make_float4(v1.x * qscale, v1.y * qscale, v1.z * qscale, v1.w * qscale);
            uint2 a1, a2, a3, b1, b2, b3;
            split4(v0, a1, a2, a3);
            split4(v1, b1, b2, b3);
            qb3[ks][0] = __builtin_bit_cast(bf16x8, make_uint4(a1.x, a1.y, b1.x, b1.y));
            qb3[ks][1] = __builtin_bit_cast(bf16x8, make_uint4(a2.x, a2.y, b2.x, b2.y));
            qb3[ks][2] = __builtin_bit_cast(bf16x8, make_uint4(a3.x, a3.y, b3.x, b3.y));
        }
    } else {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            if (q < Tq) {
                float4 v = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + kg * 8 + hi * 4);
                qf[kg] = make_float4(v.x * qscale, v.y * qscale, v.z * qscale, v.w * qscale);
            } else {
                qf[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    float4 kreg[SLOTS], vreg[SLOTS];
    auto load_tile_into = [&](int tile, float4 (&kreg)[SLOTS], float4 (&vreg)[SLOTS]) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * NTHR;
            const int key = tile * KT + f / (CH / 4), c = (f % (CH / 4)) * 4;
            if (f < KV4 && key < T) {
                float4 kv = *reinterpret_cast<const float4*>(kbase + (size_t)key * ldkv + c);
                kreg[s] = make_float4(kv.x * scale, kv.y * scale, kv.z * scale, kv.w * scale);
                vreg[s] = *reinterpret_cast<const float4*>(vbase + (size_t)key * ldkv + c);
            } else {
                kreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
                vreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto load_tile = [&](int tile) { load_tile_into(tile, kreg, vreg); };
    // one slot (a float4 of K and one of V per thread) of a staged tile: split + LDS stores
    auto store_slot = [&](int buf, int s, const float4 (&kreg)[SLOTS], const float4 (&vreg)[SLOTS]) {
        {
            const int f = tid + s * NTHR;
            if (f < KV4) {
                const int key = f / (CH / 4), c = (f % (CH / 4)) * 4;
                if (BQ) {
                    uint2 p1, p2, p3;
                    split4(kreg[s], p1, p2, p3);
                    unsigned char* kd = reinterpret_cast<unsigned char*>(kbuf + buf * KSTAGE) + key * KROWB + c * 2;
                    *reinterpret_cast<uint2*>(kd) = p1;
                    *reinterpret_cast<uint2*>(kd + KPLANE) = p2;
                    *reinterpret_cast<uint2*>(kd + 2 * KPLANE) = p3;
                } else {
                    *reinterpret_cast<float4*>(kbuf + buf * KSTAGE + key * KPITCH + c) = kreg[s];
                }
                if (BV) {
                    uint2 p1, p2, p3;
                    split4(vreg[s], p1, p2, p3);
                    unsigned char* vd = reinterpret_cast<unsigned char*>(vbuf + buf * VSTAGE) + (c >> 5) * VCT + ((c >> 4) & 1) * VSUB + key * 32 +
                                        (c & 15) * 2;
                    *reinterpret_cast<uint2*>(vd) = p1;
                    *reinterpret_cast<uint2*>(vd + VPLANE) = p2;
                    *reinterpret_cast<uint2*>(vd + 2 * VPLANE) = p3;
                } else {
                    *reinterpret_cast<float4*>(vbuf + buf * VSTAGE + key * VPITCH + c) = vreg[s];
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) store_slot(buf, s, kreg, vreg);
    };
    // PIPE: the K (part 0) or V (part 1) half of slot s of the tile held in kreg / vreg, in SIX stages of 3 - 8 instructions, one behind
    // each of the six MFMAs of a K-step (the caller fences every stage with sched_barrier: source order IS issue order)
    uint2 hp1, hp2, hp3;
    float hr0, hr1;
    auto store_half_stage = [&](int buf, int s, int part, int stage) {
        const float4 a = part == 0 ? kreg[s] : vreg[s];
        if (stage == 0) split2_a(a.x, a.y, hp1.x, hr0, hr1);
        else if (stage == 1) split2_b(hr0, hr1, hp2.x, hp3.x);
        else if (stage == 2) split2_a(a.z, a.w, hp1.y, hr0, hr1);
        else if (stage == 3) split2_b(hr0, hr1, hp2.y, hp3.y);
        else if (stage == 5) {
            const int f = tid + s * NTHR;
            const int key = f / (CH / 4), c = (f % (CH / 4)) * 4;
            if (part == 0) {
                unsigned char* kd = reinterpret_cast<unsigned char*>(kbuf + buf * KSTAGE) + key * KROWB + c * 2;
                *reinterpret_cast<uint2*>(kd) = hp1;
                *reinterpret_cast<uint2*>(kd + KPLANE) = hp2;
                *reinterpret_cast<uint2*>(kd + 2 * KPLANE) = hp3;
            } else {
                unsigned char* vd = reinterpret_cast<unsigned char*>(vbuf + buf * VSTAGE) + (c >> 5) * VCT + ((c >> 4) & 1) * VSUB + key * 32 +
                                    (c & 15) * 2;
                *reinterpret_cast<uint2*>(vd) = hp1;
                *reinterpret_cast<uint2*>(vd + VPLANE) = hp2;
                *reinterpret_cast<uint2*>(vd + 2 * VPLANE) = hp3;
            }
        }
    };

    if (CH < 32) {   // rows of V^T beyond CH are read by the MFMA A operand: keep them finite
        for (int i = tid; i < 2 * VSTAGE; i += NTHR) vbuf[i] = 0.f;
        __syncthreads();
    }

    const int ntiles = (T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    if (PIPE) load_tile(1);                        // (rows beyond T load as zeros: harmless when there is no tile 1)
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (!PIPE && tile + 1 < ntiles) load_tile(tile + 1);     // (PIPE: kreg / vreg already hold tile + 1, requested a tile ago)

        // ---- S^T = K Q^T ------------------------------------------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if (BQ) {
            // one accumulator: a chain of MFMAs into the same registers issues back to back on this part (measured against two
            // interleaved accumulators + 16 adds per key tile: 6.21 -> 6.13 ms at T = 4096)
            const unsigned char* kb = reinterpret_cast<const unsigned char*>(kbuf + buf * KSTAGE) + lq * KROWB + hi * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const bf16x8*>(kb + p * KPLANE + ks * 32);
                if constexpr (PIPE) {
                    // ... with stage t of piece ks of the NEXT tile's staging behind MFMA t (K of slot ks / 2, then its V: 2 SLOTS == KS
                    // pieces; unconditional -- behind the last tile it rewrites a stage nobody reads again -- so that it shares the
                    // basic block of the MFMAs).  The MFMAs of a K-step form a dependent chain: the wave would otherwise sit out ~30
                    // cycles behind each of them with its split still to do.
                    static_assert(!PIPE || 2 * SLOTS == KS, "one staging piece per K-step");
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[BF3_TA[t]], qb3[ks][BF3_TB[t]], s, 0, 0, 0);
                        store_half_stage(buf ^ 1, ks >> 1, ks & 1, t);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 6; ++t) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[BF3_TA[t]], qb3[ks][BF3_TB[t]], s, 0, 0, 0);
                }
            }
        } else {
            const float* kb = kbuf + buf * KSTAGE + lq * KPITCH + hi * 4;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const float4 kf = *reinterpret_cast<const float4*>(kb + kg * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kg].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kg].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kg].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kg].w, s, 0, 0, 0);
            }
        }

        if (PIPE) load_tile(tile + 2);      // the staging registers are free again: request the tile after next (rows beyond T: zeros);
                                            // it lands under the softmax and the P V product of this tile
        // ---- online softmax over the 32 keys of this tile (16 here, 16 in lane^32) --------------------------------
        // (scores carry the factor log2 e -- folded into the q scale above -- so that the exponentials are bare v_exp_f32; keys
        // beyond T exist only in the last tile: the 32 compare / select pairs stay out of every other iteration)
        if (tile * KT + KT > T) {
            const int key0 = tile * KT + 4 * hi;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) >= T) s[r] = -INFINITY;
            BBDM_KEEP_IN_BRANCH(s);                  // (if-converted, the 32 selects would run in every iteration again)
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        const float m_run_prev = m_run;
        m_run = m_new;
        if (__ballot(m_new != m_run_prev) != 0ull) {   // the running maximum rarely moves after the first tiles: no rescale then (alpha == 1)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
                BBDM_KEEP_IN_BRANCH(o[ct]);
            }
        }

        // ---- O^T += V^T P^T ------------------------------------------------------------------------------------
        if (BV) {
            typedef short v4s __attribute__((ext_vector_type(4)));
            // transpose-read address of this lane: 16-lane group g reads subtile 2 ct + g; lane i of the group supplies the 8-byte
            // word (key row i >> 2, channel quad i & 3) and receives channel i, 4 consecutive keys
            const unsigned char* vt = reinterpret_cast<const unsigned char*>(vbuf + buf * VSTAGE) + ((lane >> 4) & 1) * VSUB +
                                      (4 * hi + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
            bf16x8 pfs[2][3];                                   // P^T planes of the lane's 8 k-slots of K-step ks: registers 8 ks .. 8 ks + 7
            unsigned pw[3][4];
            float pr0, pr1;
            auto split_p = [&](int ks) {
#pragma unroll
                for (int d = 0; d < 4; ++d) split2(s[8 * ks + 2 * d], s[8 * ks + 2 * d + 1], pw[0][d], pw[1][d], pw[2][d]);
#pragma unroll
                for (int p = 0; p < 3; ++p) pfs[ks][p] = __builtin_bit_cast(bf16x8, make_uint4(pw[p][0], pw[p][1], pw[p][2], pw[p][3]));
            };
            // PIPE: the split of key half 1 in eight stages behind the first eight MFMAs of key half 0
            auto split_p1_stage = [&](int j) {
                if (j < 8) {
                    const int d = j >> 1;
                    if ((j & 1) == 0) split2_a(s[8 + 2 * d], s[8 + 2 * d + 1], pw[0][d], pr0, pr1);
                    else split2_b(pr0, pr1, pw[1][d], pw[2][d]);
                }
                if (j == 8)
#pragma unroll
                    for (int p = 0; p < 3; ++p) pfs[1][p] = __builtin_bit_cast(bf16x8, make_uint4(pw[p][0], pw[p][1], pw[p][2], pw[p][3]));
            };
            if (PIPE) split_p(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (!PIPE) split_p(ks);
                const bf16x8 (&pf)[3] = pfs[ks];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    bf16x8 vf[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned char* a = vt + p * VPLANE + ct * VCT + ks * 16 * 32;
                        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a));
                        const v4s hi8 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(a + 8 * 32));
                        typedef short v8s __attribute__((ext_vector_type(8)));
                        const v8s both = {lo[0], lo[1], lo[2], lo[3], hi8[0], hi8[1], hi8[2], hi8[3]};
                        vf[p] = __builtin_bit_cast(bf16x8, both);
                    }
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[BF3_TA[t]], pf[BF3_TB[t]], o[ct], 0, 0, 0);
                        if (PIPE && ks == 0) {
                            split_p1_stage(ct * 6 + t);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (PIPE && ks == 0 && 6 * CT < 9) {            // (one channel tile: the stages that found no MFMA)
#pragma unroll
                    for (int j = 6 * CT; j < 9; ++j) split_p1_stage(j);
                }
            }
        } else {
            const float* vb = vbuf + buf * VSTAGE + lq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float vf = vb[krow * VPITCH + ct * 32];
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], o[ct], 0, 0, 0);
                }
            }
        }

        if (!PIPE && tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O^T / l -> LDS [query][c] -> coalesced rows ----------------------------------------------------
    float* obuf = smem;
    const float inv = 1.0f / l_run;
    if (lse && hi == 0 && q < Tq) lse[((size_t)n * heads + h) * Tq + q] = m_run * LN2 + logf(l_run);   // for the backward pass (natural log)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (c < CH) obuf[(wave * 32 + lq) * OPITCH + c] = o[ct][r] * inv;
        }
    __syncthreads();
    for (int i = tid; i < QB * CH; i += NTHR) {
        const int ql = i / CH, c = i - ql * CH;
        const int qq = qb * QB + ql;
        if (qq < Tq) out[((size_t)n * Tq + qq) * ldo + h * CH + c] = obuf[ql * OPITCH + c];
    }
}

// ---- K / V pre-split (round 5): the operand planes of the key tiles are written ONCE per (image, head) instead of being split by every
// 128-query workgroup that walks them (at T = 4096: 32 times) ---------------------------------------------------------------------------
// With the split in the main loop the kernel issued 6.5 VALU instructions per MFMA and 32 % of its LDS cycles were bank conflicts of
// the 8-byte plane stores (profiles/r05_pmc_c2_counters.md); a third of both was the K / V staging.  attn_kv_planes_kernel writes, per
// (image, head) and 32-key tile, 24 "fragment units" of 1 KB -- the operand layout of gemm_bf3p.hip: a unit is one bf16 plane of a
// 32-row x 16-k MFMA A operand in the order the lanes consume it, element (row, k) at byte (k >> 3) * 512 + row * 16 + (k & 7) * 2 --
//   units  0 .. 11: K  [ks = 16-channel step][plane]           rows = keys, k = channels 16 ks ..        (scaled like the staged K)
//   units 12 .. 23: V^T[ct = 32-channel tile][ks2][plane]      rows = channels, k-slot (half, j) = key 16 ks2 + 4 half + j (j < 4),
//                                                              16 ks2 + 8 + 4 half + (j - 4) otherwise: the keys register
//                                                              8 ks2 + j of the in-place P^T operand holds (see attn_fwd_kernel)
// and attn_fwd_planes_kernel copies a tile's 24 KB into LDS by LDS-DMA (6 copies per wave, a tile ahead), reads every fragment with
// one linear ds_read_b128 at lane * 16 (conflict-free by construction) and spends its VALU on the softmax and the P split alone.
// Measured at C2 (N 16, T 4096, 16 x 64; tools/attn_bench.py, ablated builds): one launch 5.49 ms; this pair 5.06 + 0.26 ms.  Of the 5.06:
// MFMAs alone 3.82 (1.73 PFLOP/s of bf16), + softmax / P split 0.7, + copies and the per-tile barrier 0.45, fragment reads 0 -- they add
// up.  A touch of the planes three tiles ahead (4-byte LDS-DMAs into a junk kilobyte, to have the lines in L2) cost 0.22 ms and is gone.
// Same split, same six terms, same MFMA order as attn_fwd_kernel<CH, true, true>: bit-equal results (tests).
constexpr int ATTN_UNIT = 1024;

// NP = 2 (round 6): the fp16-pair planes of h2_split.h under ONE power-of-two scale 2^e for Q, K and V, e from `bound` >= max |qkv|
// (the planner's provable bound of the qkv projection: GroupNorm bound x max row L1 of the weight + max |bias|; the softmax scale
// ch^-1/4 only shrinks q and k): NP units per (step, operand) instead of three, three MFMA terms instead of six.
template <int CH, int NP = 3>
__global__ void __launch_bounds__(256) attn_kv_planes_kernel(const float* __restrict__ ksrc, const float* __restrict__ vsrc, int ldkv, int hskv,
                                                             unsigned char* __restrict__ planes, int T, int heads, float scale,
                                                             const float* __restrict__ bound) {
    constexpr int KS = CH / 16, CT = CH / 32, UNITS = NP * KS + 2 * NP * CT, PITCH = CH + 4;
    __shared__ __attribute__((aligned(16))) float kt[KT * PITCH], vt[KT * PITCH];
    const int ntiles = T / KT;
    const int tile = (int)(blockIdx.x % (unsigned)ntiles), nh = (int)(blockIdx.x / (unsigned)ntiles);
    const int h = nh % heads, n = nh / heads;
    const float* kbase = ksrc + ((size_t)n * T + (size_t)tile * KT) * ldkv + h * hskv;
    const float* vbase = vsrc + ((size_t)n * T + (size_t)tile * KT) * ldkv + h * hskv;
    const int tid = threadIdx.x, lane = tid & 63;
    float up = 1.f;                                                // 2^e (exact multiplications)
    if constexpr (NP == 2) up = h2_pow2(h2_exp_of_bound(*bound));
    for (int f = tid; f < KT * CH / 4; f += 256) {
        const int key = f / (CH / 4), c = (f % (CH / 4)) * 4;
        const float4 kv = *reinterpret_cast<const float4*>(kbase + (size_t)key * ldkv + c);
        const float4 vv = *reinterpret_cast<const float4*>(vbase + (size_t)key * ldkv + c);
        *reinterpret_cast<float4*>(kt + key * PITCH + c) = make_float4(kv.x * scale * up, kv.y * scale * up, kv.z * scale * up, kv.w * scale * up);
        *reinterpret_cast<float4*>(vt + key * PITCH + c) = make_float4(vv.x * up, vv.y * up, vv.z * up, vv.w * up);
    }
    __syncthreads();
    unsigned char* dst = planes + ((size_t)nh * ntiles + tile) * (size_t)(UNITS * ATTN_UNIT) + lane * 16;
    const int row = lane & 31, half = lane >> 5;
    for (int item = tid >> 6; item < KS + 2 * CT; item += 4) {
        float4 v0, v1;
        int u;
        if (item < KS) {                          // K, step ks = item: key `row`, channels 16 ks + 8 half ..
            const float* src = kt + row * PITCH + item * 16 + half * 8;
            v0 = *reinterpret_cast<const float4*>(src);
            v1 = *reinterpret_cast<const float4*>(src + 4);
            u = item * NP;
        } else {                                  // V^T, (ct, ks2): channel 32 ct + row, the 8 keys of this lane's k-slots
            const int ct = (item - KS) >> 1, ks2 = (item - KS) & 1;
            const float* src = vt + (16 * ks2 + 4 * half) * PITCH + ct * 32 + row;
            v0 = make_float4(src[0], src[PITCH], src[2 * PITCH], src[3 * PITCH]);
            v1 = make_float4(src[8 * PITCH], src[9 * PITCH], src[10 * PITCH], src[11 * PITCH]);
            u = NP * KS + ((ct * 2 + ks2) * NP);
        }
        if constexpr (NP == 3) {
            uint2 a1, a2, a3, b1, b2, b3;
            split4(v0, a1, a2, a3);
            split4(v1, b1, b2, b3);
            *reinterpret_cast<uint4*>(dst + (size_t)(u + 0) * ATTN_UNIT) = make_uint4(a1.x, a1.y, b1.x, b1.y);
            *reinterpret_cast<uint4*>(dst + (size_t)(u + 1) * ATTN_UNIT) = make_uint4(a2.x, a2.y, b2.x, b2.y);
            *reinterpret_cast<uint4*>(dst + (size_t)(u + 2) * ATTN_UNIT) = make_uint4(a3.x, a3.y, b3.x, b3.y);
        } else {
            uint2 a1, a2, b1, b2;
            h2_split4(v0, a1, a2);
            h2_split4(v1, b1, b2);
            *reinterpret_cast<uint4*>(dst + (size_t)(u + 0) * ATTN_UNIT) = make_uint4(a1.x, a1.y, b1.x, b1.y);
            *reinterpret_cast<uint4*>(dst + (size_t)(u + 1) * ATTN_UNIT) = make_uint4(a2.x, a2.y, b2.x, b2.y);
        }
    }
}

// NP = 2: Q (x ch^-1/4 log2 e) is scaled by the planes' 2^e and split into two fp16 halves, S = (three terms) 2^-2e; P = exp2(S - m) in
// (0, 1] is split under ITS exact bound 1 (P 2^14 = h1 + h2); O accumulates P 2^14 V 2^e and is re-scaled once in the epilogue.  The term
// order is gemm_bf3p.hip's (h1 k2) (h1 k1) (h2 k1).
constexpr int H2_TA[3] = {0, 0, 1}, H2_TB[3] = {1, 0, 0};
template <int CH, int NP = 3>
__global__ void __launch_bounds__(256, NP == 2 ? 4 : 3) attn_fwd_planes_kernel(const float* __restrict__ qsrc, int ldq, int hsq,
                                                                  const unsigned char* __restrict__ planes, float* __restrict__ out, int ldo,
                                                                  float* __restrict__ lse, int T, int heads, int nheads_total, float qscale,
                                                                  const float* __restrict__ bound) {
    constexpr int NW = 4, QB = NW * 32, NTHR = NW * 64;
    constexpr int KS = CH / 16, CT = CH / 32, UNITS = NP * KS + 2 * NP * CT, NT = NP == 3 ? 6 : 3;
    constexpr int STAGE = UNITS * ATTN_UNIT;                       // bytes per key tile
    constexpr int OPITCH = CH + 1;
    static_assert(UNITS % NW == 0, "the tile's units are dealt evenly to the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // attn_planes_lds(CH, NP) bytes: two tile stages / the epilogue's O^T

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    const int qblocks = T / QB;
    qscale *= LOG2E;
    const int L = (int)blockIdx.x, slot = L >> 3;                  // XCD L % 8 owns the (image, head) pairs x, x + 8, ... (see attn_fwd_kernel)
    const int qb = slot % qblocks;
    const int nh = (L & 7) + 8 * (slot / qblocks);
    if (nh >= nheads_total) return;
    const int h = nh % heads, n = nh / heads;
    const float* qbase = qsrc + (size_t)n * T * ldq + h * hsq;
    const int ntiles = T / KT;
    const unsigned char* ptile = planes + (size_t)nh * ntiles * (size_t)STAGE;
    // fp16 pair: 2^e of the planes, 2^-2e for the scores, 2^14 for P (h2_exp_of_bound(1.f)), 2^-(14 + e) for O
    float up = 1.f, sdown = 1.f, odown = 1.f;
    constexpr float PUP = 16384.f;
    if constexpr (NP == 2) {
        const int e = h2_exp_of_bound(*bound);
        up = h2_pow2(e);
        sdown = h2_pow2(-e) * h2_pow2(-e);
        odown = h2_pow2(-e) * (1.f / PUP);
    }
    auto issue = [&](int tile) {                                   // this wave's share of tile's units -> stage tile & 1
        const unsigned char* src = ptile + (size_t)tile * STAGE;
        unsigned char* dstl = smem + (tile & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < UNITS / NW; ++i) {
            const int u = wave * (UNITS / NW) + i;
            glds16(src + u * ATTN_UNIT, (unsigned)lane * 16, dstl + u * ATTN_UNIT);
        }
    };
    issue(0);

    const int q = qb * QB + wave * 32 + lq;
    uint4 qf[KS][NP];                                              // lane holds q[c], c = ks*16 + hi*8 + 0..7, as NP 16-bit planes
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float4 v0 = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + ks * 16 + hi * 8);
        float4 v1 = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + ks * 16 + hi * 8 + 4);
        v0 = make_float4(v0.x * qscale * up, v0.y * qscale * up, v0.z * qscale * up, v0.w * qscale * up);
        v1 = make_float4(v1.x * qscale * up, v1.y * qscale * up, v1.z * qscale * up, v1.w * qscale * up);
        if constexpr (NP == 3) {
            uint2 a1, a2, a3, b1, b2, b3;
            split4(v0, a1, a2, a3);
            split4(v1, b1, b2, b3);
            qf[ks][0] = make_uint4(a1.x, a1.y, b1.x, b1.y);
            qf[ks][1] = make_uint4(a2.x, a2.y, b2.x, b2.y);
            qf[ks][NP - 1] = make_uint4(a3.x, a3.y, b3.x, b3.y);
        } else {
            uint2 a1, a2, b1, b2;
            h2_split4(v0, a1, a2);
            h2_split4(v1, b1, b2);
            qf[ks][0] = make_uint4(a1.x, a1.y, b1.x, b1.y);
            qf[ks][1] = make_uint4(a2.x, a2.y, b2.x, b2.y);
        }
    }
#define ATTN_MFMA(A_, B_, C_)                                                                                                    \
    (NP == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)   \
             : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, B_), C_, 0, 0, 0))
    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    wait_vmcnt<0>();
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        if (tile + 1 < ntiles) issue(tile + 1);                    // (its stage was last read a tile ago: the barrier below is behind)
        const unsigned char* st = smem + (tile & 1) * STAGE + lane * 16;
        // ---- S^T = K Q^T: one chain of NT KS MFMAs (see attn_fwd_kernel) ---------------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 kf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) kf[p] = *reinterpret_cast<const uint4*>(st + (ks * NP + p) * ATTN_UNIT);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                s = ATTN_MFMA(kf[NP == 3 ? BF3_TA[t] : H2_TA[t % 3]], qf[ks][NP == 3 ? BF3_TB[t] : H2_TB[t % 3]], s);
        }
        // ---- online softmax (base 2; no keys beyond T: T is a multiple of the tile) ------------------------------------------------
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
        if constexpr (NP == 2) mt *= sdown;                        // (an exact power of two: max and scaling commute)
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(NP == 2 ? __builtin_fmaf(s[r], sdown, -m_new) : s[r] - m_new);     // (s 2^-2e is exact: one rounding either way)
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        const float m_run_prev = m_run;
        m_run = m_new;
        if (__ballot(m_new != m_run_prev) != 0ull) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
                BBDM_KEEP_IN_BRANCH(o[ct]);
            }
        }
        // ---- O^T += V^T P^T: P^T in place (registers 8 ks2 .. 8 ks2 + 7 are the lane's k-slots of step ks2); the split of key half 1
        // is dealt between the first MFMAs of half 0 ---------------------------------------------------------------------------------
        uint4 pfs[2][NP];
        unsigned pw[NP][4];
        float pr0, pr1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if constexpr (NP == 3) split2(s[2 * d], s[2 * d + 1], pw[0][d], pw[1][d], pw[NP - 1][d]);
            else h2_split2(s[2 * d] * PUP, s[2 * d + 1] * PUP, pw[0][d], pw[1][d]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) pfs[0][p] = make_uint4(pw[p][0], pw[p][1], pw[p][2], pw[p][3]);
        auto split_p1_stage = [&](int j) {
            if (j < 8) {
                const int d = j >> 1;
                if constexpr (NP == 3) {
                    if ((j & 1) == 0) split2_a(s[8 + 2 * d], s[8 + 2 * d + 1], pw[0][d], pr0, pr1);
                    else split2_b(pr0, pr1, pw[1][d], pw[NP - 1][d]);
                } else {
                    if ((j & 1) == 0) {
                        const float x0 = s[8 + 2 * d] * PUP, x1 = s[8 + 2 * d + 1] * PUP;
                        pw[0][d] = cvt_pk_h(x0, x1);
                        const f16x2 hh = __builtin_bit_cast(f16x2, pw[0][d]);
                        pr0 = x0 - (float)hh.x;
                        pr1 = x1 - (float)hh.y;
                    } else
                        pw[1][d] = cvt_pk_h(pr0, pr1);
                }
            }
            if (j == 8)
#pragma unroll
                for (int p = 0; p < NP; ++p) pfs[1][p] = make_uint4(pw[p][0], pw[p][1], pw[p][2], pw[p][3]);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 (&pf)[NP] = pfs[ks];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint4 vf[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) vf[p] = *reinterpret_cast<const uint4*>(st + (NP * KS + (ct * 2 + ks) * NP + p) * ATTN_UNIT);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    o[ct] = ATTN_MFMA(vf[NP == 3 ? BF3_TA[t] : H2_TA[t % 3]], pf[NP == 3 ? BF3_TB[t] : H2_TB[t % 3]], o[ct]);
                    if (ks == 0) {
                        split_p1_stage(ct * NT + t);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (ks == 0 && NT * CT < 9) {
#pragma unroll
                for (int j = NT * CT; j < 9; ++j) split_p1_stage(j);
            }
        }
        wait_vmcnt<0>();                         // my copies of tile + 1 have landed ...
        __syncthreads();                         // ... everybody's, and everybody is done with this tile's stage
    }
#undef ATTN_MFMA

    // ---- epilogue: O^T / l -> LDS [query][c] -> coalesced rows (as attn_fwd_kernel) -----------------------------------------------------
    float* obuf = reinterpret_cast<float*>(smem);
    const float inv = (NP == 2 ? odown : 1.0f) / l_run;
    if (lse && hi == 0) lse[((size_t)n * heads + h) * T + q] = m_run * LN2 + logf(l_run);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            obuf[(wave * 32 + lq) * OPITCH + c] = o[ct][r] * inv;
        }
    __syncthreads();
    for (int i = tid; i < QB * CH; i += NTHR) {
        const int ql = i / CH, c = i - ql * CH;
        out[((size_t)n * T + qb * QB + ql) * ldo + h * CH + c] = obuf[ql * OPITCH + c];
    }
}

}  // namespace

static int launch_attention(const float* q, int ldq, int hsq, const float* k, const float* v, int ldkv, int hskv, float* out,
                            int ldo, float* lse, int N, int Tq, int Tk, int heads, int ch, float qscale, float kscale,
                            hipStream_t st) {
    // Q K^T and P V on the bf16x3 path (option attn_bf3 = 0: both on the f32 MFMA; 2: only Q K^T, the round-2 kernel -- for the A/B)
    const int bq = bbdm_option(BBDM_OPT_ATTN_BF3);
    const bool pipe = bbdm_option(BBDM_OPT_ATTN_PIPE) != 0;           // the interleaved main loop (PIPE) of the full bf16x3 path (2: + the pre-split entry points)
    // 4 waves (128 queries) per workgroup, three workgroups per CU (an 8-wave / 256-query form, K / V staged once per 256 queries,
    // measured 6.55 against 6.16 ms at C2 in round 3 and is gone)
    const int nw = 4;
    const int qblocks = (Tq + nw * 32 - 1) / (nw * 32);
    const int nht = N * heads;
    const dim3 grid((unsigned)(8ll * ((nht + 7) / 8) * qblocks));
#define BBDM_ATTN_FWD(CH, BQ, BV, NW)                                                                                      \
    hipLaunchKernelGGL((attn_fwd_kernel<CH, BQ, BV, NW>), grid, dim3(NW * 64), 0, st, q, ldq, hsq, k, v, ldkv, hskv, out, ldo, \
                       lse, Tq, Tk, heads, nht, qscale, kscale)
#define BBDM_ATTN_FWD_NW(CH, BQ, BV) BBDM_ATTN_FWD(CH, BQ, BV, 4)
#define BBDM_ATTN_FWD_P(CH)                                                                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<CH, true, true, 4, true>), grid, dim3(4 * 64), 0, st, q, ldq, hsq, k, v, ldkv, hskv, out, ldo, \
                       lse, Tq, Tk, heads, nht, qscale, kscale)
    if (ch == 64) { if (bq == 1 && pipe) BBDM_ATTN_FWD_P(64); else if (bq == 1) BBDM_ATTN_FWD_NW(64, true, true); else if (bq) BBDM_ATTN_FWD_NW(64, true, false); else BBDM_ATTN_FWD_NW(64, false, false); }
    else if (ch == 32) { if (bq == 1 && pipe) BBDM_ATTN_FWD_P(32); else if (bq == 1) BBDM_ATTN_FWD_NW(32, true, true); else if (bq) BBDM_ATTN_FWD_NW(32, true, false); else BBDM_ATTN_FWD_NW(32, false, false); }
#undef BBDM_ATTN_FWD_P
    else { if (bq) BBDM_ATTN_FWD_NW(16, true, false); else BBDM_ATTN_FWD_NW(16, false, false); }
#undef BBDM_ATTN_FWD_NW
#undef BBDM_ATTN_FWD
    return 0;
}

extern "C" int bbdm_attention_f32(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads,
                                  int ch, int new_order, void* stream) {
    BBDM_REQUIRE(qkv && out, "attention: null pointer");
    BBDM_REQUIRE(N > 0 && T > 0 && heads > 0, "attention: bad shape");
    BBDM_REQUIRE(ch == 16 || ch == 32 || ch == 64, "attention: head channels %d unsupported (16, 32, 64)", ch);
    BBDM_REQUIRE(ldq % 4 == 0 && ldq >= 3 * heads * ch && ldo >= heads * ch && ((uintptr_t)qkv & 15) == 0,
                 "attention: bad pitch / alignment");
    const float scale = 1.0f / sqrtf(sqrtf((float)ch));
    const int C = heads * ch;
    if (new_order)      // QKVAttention: q | k | v thirds, head h at h * ch inside each
        launch_attention(qkv, ldq, ch, qkv + C, qkv + 2 * C, ldq, ch, out, ldo, lse, N, T, T, heads, ch, scale, scale,
                         (hipStream_t)stream);
    else                // QKVAttentionLegacy: per head (q, k, v) triples
        launch_attention(qkv, ldq, 3 * ch, qkv + ch, qkv + 2 * ch, ldq, 3 * ch, out, ldo, lse, N, T, T, heads, ch, scale,
                         scale, (hipStream_t)stream);
    BBDM_CHECK_LAUNCH("attention");
    return BBDM_OK;
}

// ---- the pre-split form (see attn_kv_planes_kernel): a launch that writes the K / V operand planes of every (image, head) and an
// attention launch that reads them.  bbdm_attention_kv_planes_bytes returns 0 where the form does not apply (head channels other than
// 64 / 32, T not a multiple of 128, sequences too short for the extra launch to pay); the results equal bbdm_attention_f32's bit for bit.
static size_t attn_planes_units(int ch, int np) { return (size_t)(np * (ch / 16) + 2 * np * (ch / 32)); }
static size_t attn_planes_lds(int ch, int np) {
    const size_t stages = 2 * attn_planes_units(ch, np) * 1024, epilogue = 128 * (size_t)(ch + 1) * 4;
    return stages > epilogue ? stages : epilogue;
}
static bool attn_planes_ok(int N, int T, int heads, int ch) {
    const int mode = bbdm_option(BBDM_OPT_ATTN_PIPE);               // 2: long sequences, 3: every shape the layout takes (tests)
    return (ch == 64 || ch == 32) && T % 128 == 0 && (T >= 1024 || mode >= 3) && bbdm_option(BBDM_OPT_ATTN_BF3) == 1 && mode >= 2;
}
extern "C" size_t bbdm_attention_kv_planes_bytes(int N, int T, int heads, int ch) {
    if (N <= 0 || T <= 0 || heads <= 0 || !attn_planes_ok(N, T, heads, ch)) return 0;
    return (size_t)N * heads * (T / 32) * attn_planes_units(ch, 3) * 1024;
}
extern "C" size_t bbdm_attention_kv_planes_h2_bytes(int N, int T, int heads, int ch) {
    if (N <= 0 || T <= 0 || heads <= 0 || !attn_planes_ok(N, T, heads, ch)) return 0;
    return (size_t)N * heads * (T / 32) * attn_planes_units(ch, 2) * 1024;
}
static int attn_kv_planes(const float* qkv, int ldq, void* planes, size_t planes_bytes, int N, int T, int heads, int ch, int new_order,
                          const float* bound, void* stream) {
    BBDM_REQUIRE(qkv && planes, "attention_kv_planes: null pointer");
    const size_t need = bound ? bbdm_attention_kv_planes_h2_bytes(N, T, heads, ch) : bbdm_attention_kv_planes_bytes(N, T, heads, ch);
    BBDM_REQUIRE(need != 0, "attention_kv_planes: N=%d T=%d heads=%d ch=%d has no pre-split form (bbdm_attention_kv_planes_bytes is 0)", N, T, heads, ch);
    BBDM_REQUIRE(planes_bytes >= need, "attention_kv_planes: %zu bytes of planes, %zu needed", planes_bytes, need);
    BBDM_REQUIRE(ldq % 4 == 0 && ldq >= 3 * heads * ch && (((uintptr_t)qkv | (uintptr_t)planes) & 15) == 0, "attention_kv_planes: bad pitch / alignment");
    BBDM_REQUIRE((long long)N * heads * (T / 32) < (1ll << 31), "attention_kv_planes: too many key tiles");
    const float scale = 1.0f / sqrtf(sqrtf((float)ch));
    const int C = heads * ch;
    const float* k = new_order ? qkv + C : qkv + ch;
    const float* v = new_order ? qkv + 2 * C : qkv + 2 * ch;
    const int hs = new_order ? ch : 3 * ch;
    const dim3 grid((unsigned)(N * heads * (T / 32)));
#define BBDM_ATTN_KV(CH, NP) hipLaunchKernelGGL((attn_kv_planes_kernel<CH, NP>), grid, dim3(256), 0, (hipStream_t)stream, k, v, ldq, hs, (unsigned char*)planes, T, heads, scale, bound)
    if (ch == 64) { if (bound) BBDM_ATTN_KV(64, 2); else BBDM_ATTN_KV(64, 3); }
    else { if (bound) BBDM_ATTN_KV(32, 2); else BBDM_ATTN_KV(32, 3); }
#undef BBDM_ATTN_KV
    BBDM_CHECK_LAUNCH("attention_kv_planes");
    return BBDM_OK;
}
static int attn_planes(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads, int ch, int new_order,
                       const void* planes, const float* bound, void* stream) {
    BBDM_REQUIRE(qkv && planes && out, "attention_planes: null pointer");
    BBDM_REQUIRE(bbdm_attention_kv_planes_bytes(N, T, heads, ch) != 0, "attention_planes: N=%d T=%d heads=%d ch=%d has no pre-split form", N, T, heads, ch);
    BBDM_REQUIRE(ldq % 4 == 0 && ldq >= 3 * heads * ch && ldo >= heads * ch && (((uintptr_t)qkv | (uintptr_t)planes) & 15) == 0,
                 "attention_planes: bad pitch / alignment");
    const float scale = 1.0f / sqrtf(sqrtf((float)ch));
    const int nht = N * heads, qblocks = T / 128;
    const dim3 grid((unsigned)(8ll * ((nht + 7) / 8) * qblocks));
    const int hsq = new_order ? ch : 3 * ch;
#define BBDM_ATTN_PL(CH, NP) hipLaunchKernelGGL((attn_fwd_planes_kernel<CH, NP>), grid, dim3(256), attn_planes_lds(CH, NP), (hipStream_t)stream, qkv, ldq, hsq, (const unsigned char*)planes, out, ldo, lse, T, heads, nht, scale, bound)
    if (ch == 64) { if (bound) BBDM_ATTN_PL(64, 2); else BBDM_ATTN_PL(64, 3); }
    else { if (bound) BBDM_ATTN_PL(32, 2); else BBDM_ATTN_PL(32, 3); }
#undef BBDM_ATTN_PL
    BBDM_CHECK_LAUNCH("attention_planes");
    return BBDM_OK;
}
extern "C" int bbdm_attention_kv_planes_f32(const float* qkv, int ldq, void* planes, size_t planes_bytes, int N, int T, int heads, int ch,
                                            int new_order, void* stream) {
    return attn_kv_planes(qkv, ldq, planes, planes_bytes, N, T, heads, ch, new_order, nullptr, stream);
}
extern "C" int bbdm_attention_planes_f32(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads, int ch,
                                         int new_order, const void* planes, void* stream) {
    return attn_planes(qkv, ldq, out, ldo, lse, N, T, heads, ch, new_order, planes, nullptr, stream);
}
// ... on the fp16-pair planes (round 6): `bound` = a device float >= max |qkv| (see attn_kv_planes_kernel); both launches take the same one.
extern "C" int bbdm_attention_kv_planes_h2_f32(const float* qkv, int ldq, void* planes, size_t planes_bytes, int N, int T, int heads, int ch,
                                               int new_order, const float* bound, void* stream) {
    BBDM_REQUIRE(bound, "attention_kv_planes_h2: null bound");
    return attn_kv_planes(qkv, ldq, planes, planes_bytes, N, T, heads, ch, new_order, bound, stream);
}
extern "C" int bbdm_attention_planes_h2_f32(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads, int ch,
                                            int new_order, const void* planes, const float* bound, void* stream) {
    BBDM_REQUIRE(bound, "attention_planes_h2: null bound");
    return attn_planes(qkv, ldq, out, ldo, lse, N, T, heads, ch, new_order, planes, bound, stream);
}

// CrossAttention.forward (model/BrownianBridge/base/modules/attention.py:170-194): q [N][Tq][heads*ch] from the image
// tokens, k / v [N][Tk][heads*ch] from the context tokens (Tk != Tq in general; k = v source = x for self-attention),
// softmax_j(q_i . k_j * ch^-1/2) v_j, heads laid out 'b n (h d)'.
extern "C" int bbdm_cross_attention_f32(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo,
                                        float* lse, int N, int Tq, int Tk, int heads, int ch, void* stream) {
    BBDM_REQUIRE(q && k && v && out, "cross_attention: null pointer");
    BBDM_REQUIRE(N > 0 && Tq > 0 && Tk > 0 && heads > 0, "cross_attention: bad shape");
    BBDM_REQUIRE(ch == 16 || ch == 32 || ch == 64, "cross_attention: head channels %d unsupported (16, 32, 64)", ch);
    BBDM_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldq >= heads * ch && ldkv >= heads * ch && ldo >= heads * ch &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0,
                 "cross_attention: bad pitch / alignment");
    launch_attention(q, ldq, ch, k, v, ldkv, ch, out, ldo, lse, N, Tq, Tk, heads, ch, 1.0f / sqrtf((float)ch), 1.0f,
                     (hipStream_t)stream);
    BBDM_CHECK_LAUNCH("cross_attention");
    return BBDM_OK;
}
