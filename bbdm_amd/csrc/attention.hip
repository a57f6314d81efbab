// attention.hip -- fp32 QKV self-attention with a streaming (online) softmax on the f32-input matrix core.
//
// Replaces QKVAttentionLegacy.forward (openaimodel.py:359-375) and QKVAttention.forward (:398-413):
//     w = softmax_over_s( (q*s)^T (k*s) ),  a = v w^T,   s = ch^-1/4
// without ever materialising the [N*heads, T, T] score tensor (17 GB at 256^2 / batch 16).
//
// One workgroup = 4 waves = 128 consecutive queries of one (image, head); each wave owns 32 queries.
// Everything is computed TRANSPOSED so that a query is a lane (column) in every MFMA result:
//   S^T[key, query] = K[key, :] . Q^T[:, query]     A = K tile from LDS, B = Q^T held in 2*CH/4 VGPRs
//   O^T[c,   query] += V^T[c, key] . P^T[key, query] A = V^T read from LDS, B = P^T = exp(S^T - m) IN PLACE:
// the 32x32 C/D layout gives lane l the keys {(r&3)+8(r>>2)+4(l>>5)} of query l&31 in register r, and the 32x32x2
// B operand wants, for its two k, lanes <32 -> k0 and lanes >=32 -> k1.  Choosing (k0,k1) = (key_lo(r), key_lo(r)+4)
// for step r makes register r of P^T exactly the B operand of step r -- no LDS round trip, no shuffles; row
// max / sum are 15 VALU ops + one cross-half exchange.  The running max / sum / rescale are per-lane scalars.
#include "common.h"

namespace {

constexpr int KT = 32;        // keys per tile
constexpr int QB = 128;       // queries per block

template <int CH>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const float* __restrict__ qsrc, int ldq, int hsq,
                                                       const float* __restrict__ ksrc, const float* __restrict__ vsrc, int ldkv,
                                                       int hskv, float* __restrict__ out, int ldo, float* __restrict__ lse,
                                                       int Tq, int T, int heads, float qscale, float scale) {
    // q: [N][Tq][ldq], head h at channel h * hsq;  k, v: [N][T][ldkv], head h at channel h * hskv (T = number of keys).
    // qscale / scale multiply q / k while they are loaded (legacy: ch^-1/4 each; CrossAttention: ch^-1/2 and 1).
    constexpr int KPITCH = CH + 4;                 // K tile pitch: b128 reads by 32 keys conflict-free
    constexpr int VPITCH = CH < 32 ? 32 : CH;      // V tile pitch (lanes sweep channels)
    constexpr int CT = (CH + 31) / 32;             // 32-row channel tiles of O^T
    constexpr int KG = CH / 8;                     // k-groups of 8 channels for QK^T
    constexpr int KV4 = KT * CH / 4;               // float4 per K (or V) tile
    constexpr int SLOTS = (KV4 + 255) / 256;

    __shared__ __attribute__((aligned(16))) float smem[2 * KT * KPITCH + 2 * KT * VPITCH];
    float* kbuf = smem;
    float* vbuf = smem + 2 * KT * KPITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    const int qblocks = (Tq + QB - 1) / QB;
    const int qb = blockIdx.x % qblocks;
    const int nh = blockIdx.x / qblocks;
    const int h = nh % heads, n = nh / heads;
    const float* qbase = qsrc + (size_t)n * Tq * ldq + h * hsq;
    const float* kbase = ksrc + (size_t)n * T * ldkv + h * hskv;
    const float* vbase = vsrc + (size_t)n * T * ldkv + h * hskv;

    // ---- Q^T fragment: lane holds q[c] for c = kg*8 + hi*4 + j -------------------------------------------------
    const int q = qb * QB + wave * 32 + lq;
    float4 qf[KG];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        if (q < Tq) {
            float4 v = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + kg * 8 + hi * 4);
            qf[kg] = make_float4(v.x * qscale, v.y * qscale, v.z * qscale, v.w * qscale);
        } else {
            qf[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    float4 kreg[SLOTS], vreg[SLOTS];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
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
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            if (f < KV4) {
                const int key = f / (CH / 4), c = (f % (CH / 4)) * 4;
                *reinterpret_cast<float4*>(kbuf + buf * KT * KPITCH + key * KPITCH + c) = kreg[s];
                *reinterpret_cast<float4*>(vbuf + buf * KT * VPITCH + key * VPITCH + c) = vreg[s];
            }
        }
    };

    if (CH < 32) {   // rows of V^T beyond CH are read by the MFMA A operand: keep them finite
        for (int i = tid; i < 2 * KT * VPITCH; i += 256) vbuf[i] = 0.f;
        __syncthreads();
    }

    const int ntiles = (T + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);

        // ---- S^T = K Q^T ------------------------------------------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kb = kbuf + buf * KT * KPITCH + lq * KPITCH + hi * 4;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const float4 kf = *reinterpret_cast<const float4*>(kb + kg * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kg].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kg].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kg].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kg].w, s, 0, 0, 0);
        }

        // ---- online softmax over the 32 keys of this tile (16 here, 16 in lane^32) --------------------------------
        const int key0 = tile * KT + 4 * hi;
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2);
            if (key >= T) s[r] = -INFINITY;
            mt = fmaxf(mt, s[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;

        // ---- O^T += V^T P^T ------------------------------------------------------------------------------------
        const float* vb = vbuf + buf * KT * VPITCH + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float vf = vb[krow * VPITCH + ct * 32];
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], o[ct], 0, 0, 0);
            }
        }

        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O^T / l -> LDS [query][c] -> coalesced rows ----------------------------------------------------
    constexpr int OPITCH = CH + 1;
    float* obuf = smem;                                    // 128 * (CH+1) floats <= the K/V buffers for CH >= 16
    static_assert(QB * OPITCH <= 2 * KT * KPITCH + 2 * KT * VPITCH, "epilogue staging does not fit");
    const float inv = 1.0f / l_run;
    if (lse && hi == 0 && q < Tq) lse[((size_t)n * heads + h) * Tq + q] = m_run + logf(l_run);   // for the backward pass
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (c < CH) obuf[(wave * 32 + lq) * OPITCH + c] = o[ct][r] * inv;
        }
    __syncthreads();
    for (int i = tid; i < QB * CH; i += 256) {
        const int ql = i / CH, c = i - ql * CH;
        const int qq = qb * QB + ql;
        if (qq < Tq) out[((size_t)n * Tq + qq) * ldo + h * CH + c] = obuf[ql * OPITCH + c];
    }
}

}  // namespace

static int launch_attention(const float* q, int ldq, int hsq, const float* k, const float* v, int ldkv, int hskv, float* out,
                            int ldo, float* lse, int N, int Tq, int Tk, int heads, int ch, float qscale, float kscale,
                            hipStream_t st) {
    const int qblocks = (Tq + QB - 1) / QB;
    const dim3 grid((unsigned)((long long)N * heads * qblocks));
    if (ch == 64)
        hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, st, q, ldq, hsq, k, v, ldkv, hskv, out, ldo, lse, Tq, Tk,
                           heads, qscale, kscale);
    else if (ch == 32)
        hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(256), 0, st, q, ldq, hsq, k, v, ldkv, hskv, out, ldo, lse, Tq, Tk,
                           heads, qscale, kscale);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, dim3(256), 0, st, q, ldq, hsq, k, v, ldkv, hskv, out, ldo, lse, Tq, Tk,
                           heads, qscale, kscale);
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
