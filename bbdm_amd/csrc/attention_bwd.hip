// attention_bwd.hip -- backward of the streaming fp32 attention (training path).
//
// Reference: autograd through QKVAttentionLegacy / QKVAttention (openaimodel.py:359-375, 398-413); the reference
// wraps the whole AttentionBlock in CheckpointFunction (util.py:119-148, openaimodel.py:318), i.e. it re-runs the
// forward in backward -- here the probabilities are recomputed from the saved per-query log-sum-exp instead.
//
//   qs = q s, ks = k s (s = ch^-1/4),  S = qs.ks,  P = exp(S - lse),  O = P V
//   D[t]  = sum_c dO[t,c] O[t,c]                                   (prep kernel)
//   dS    = P o (dO.V^T - D),  dq = s sum_s dS ks,  dk = s sum_t dS qs,  dv = sum_t P dO
// The kernels take q / k / v / dq / dk / dv as separate pointers with their own pitches and per-head channel strides, Tq
// queries against Tk keys and separate q / k scales: the packed-qkv self-attention of the AttentionBlock (s = ch^-1/4 on both)
// and CrossAttention (attention.py:170-194: q scaled by ch^-1/2, keys / values from the context tokens, Tk != Tq) share them.
// Two kernels in the same transposed-operand style as the forward (a query / a key is a LANE, so lse and D are
// per-lane scalars and the recomputed P / dS registers are used in place as the next MFMA's B operand):
//   attn_bwd_dq  : workgroup = 128 queries, streams key tiles   -> dq
//   attn_bwd_dkv : workgroup = 128 keys,    streams query tiles -> dk, dv
#include "common.h"

namespace {

constexpr int TT = 32;         // streamed tile (keys in dq, queries in dkv)
constexpr int BB = 128;        // rows owned by a block

// D[n][h][t] = sum_c dO[n,t,h*CH+c] * O[n,t,h*CH+c]
__global__ void attn_bwd_prep_kernel(const float* __restrict__ out, int ldo, const float* __restrict__ dout, int lddo,
                                     float* __restrict__ D, int N, int T, int heads, int CH) {
    const long long total = (long long)N * heads * T;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const long long nh = i / T;
        const int h = (int)(nh % heads), n = (int)(nh / heads);
        const float* o = out + ((size_t)n * T + t) * ldo + h * CH;
        const float* d = dout + ((size_t)n * T + t) * lddo + h * CH;
        float s = 0.f;
        for (int c = 0; c < CH; c += 4) {
            const float4 a = *reinterpret_cast<const float4*>(o + c), b = *reinterpret_cast<const float4*>(d + c);
            s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
        D[i] = s;
    }
}

// Store a transposed accumulator tile set (rows = channel, lane = row index of the block) through LDS.
template <int CH, int CT>
__device__ __forceinline__ void store_rows(float* stage, const f32x16 (&acc)[CT], float mul, float* __restrict__ dst,
                                           int lddst, int row0, int T, int wave, int lq, int hi, int tid) {
    constexpr int OP = CH + 1;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (c < CH) stage[(wave * 32 + lq) * OP + c] = acc[ct][r] * mul;
        }
    __syncthreads();
    for (int i = tid; i < BB * CH; i += 256) {
        const int rl = i / CH, c = i - rl * CH;
        if (row0 + rl < T) dst[(size_t)(row0 + rl) * lddst + c] = stage[rl * OP + c];
    }
    __syncthreads();
}

template <int CH>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const float* __restrict__ qsrc, int ldq, int hsq,
                                                          const float* __restrict__ ksrc, const float* __restrict__ vsrc,
                                                          int ldkv, int hskv, const float* __restrict__ dout, int lddo,
                                                          const float* __restrict__ lse, const float* __restrict__ Dv,
                                                          float* __restrict__ dqdst, int lddq, int Tq, int T, int heads,
                                                          float qscale, float scale) {
    // T = number of keys; q / dq rows [N][Tq], head h at channel h * hsq; k, v rows [N][T], head h at channel h * hskv
    constexpr int KP = CH + 4;
    constexpr int CT = (CH + 31) / 32;
    constexpr int KG = CH / 8;
    constexpr int KV4 = TT * CH / 4;
    constexpr int SLOTS = (KV4 + 255) / 256;
    constexpr int TILE = TT * KP;
    constexpr int STAGE = BB * (CH + 1);
    constexpr int SMEM = (4 * TILE > STAGE ? 4 * TILE : STAGE) + 64;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* kbuf = smem;                 // [2][TT][KP]
    float* vbuf = smem + 2 * TILE;      // [2][TT][KP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, lq = lane & 31;
    const int qblocks = (Tq + BB - 1) / BB;
    const int qb = blockIdx.x % qblocks, nh = blockIdx.x / qblocks, h = nh % heads, n = nh / heads;
    const float* qbase = qsrc + (size_t)n * Tq * ldq + h * hsq;
    const float* kbase = ksrc + (size_t)n * T * ldkv + h * hskv;
    const float* vbase = vsrc + (size_t)n * T * ldkv + h * hskv;
    const float* dob = dout + (size_t)n * Tq * lddo + h * CH;
    const int q = qb * BB + wave * 32 + lq;
    const bool qok = q < Tq;

    float4 qf[KG], dof[KG];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        qf[kg] = dof[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qok) {
            const float4 v = *reinterpret_cast<const float4*>(qbase + (size_t)q * ldq + kg * 8 + hi * 4);
            qf[kg] = make_float4(v.x * qscale, v.y * qscale, v.z * qscale, v.w * qscale);
            dof[kg] = *reinterpret_cast<const float4*>(dob + (size_t)q * lddo + kg * 8 + hi * 4);
        }
    }
    const float Lq = qok ? lse[((size_t)n * heads + h) * Tq + q] : 0.f;
    const float Dq = qok ? Dv[((size_t)n * heads + h) * Tq + q] : 0.f;

    f32x16 dq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[ct][r] = 0.f;

    float4 kreg[SLOTS], vreg[SLOTS];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            const int key = tile * TT + f / (CH / 4), c = (f % (CH / 4)) * 4;
            kreg[s] = vreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < KV4 && key < T) {
                const float4 kv = *reinterpret_cast<const float4*>(kbase + (size_t)key * ldkv + c);
                kreg[s] = make_float4(kv.x * scale, kv.y * scale, kv.z * scale, kv.w * scale);
                vreg[s] = *reinterpret_cast<const float4*>(vbase + (size_t)key * ldkv + c);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            if (f < KV4) {
                const int key = f / (CH / 4), c = (f % (CH / 4)) * 4;
                *reinterpret_cast<float4*>(kbuf + buf * TILE + key * KP + c) = kreg[s];
                *reinterpret_cast<float4*>(vbuf + buf * TILE + key * KP + c) = vreg[s];
            }
        }
    };
    for (int i = tid; i < SMEM; i += 256) smem[i] = 0.f;      // pads / rows >= CH read by A operands stay finite
    __syncthreads();
    const int ntiles = (T + TT - 1) / TT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        const float* kb = kbuf + buf * TILE + lq * KP + hi * 4;
        const float* vb = vbuf + buf * TILE + lq * KP + hi * 4;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const float4 kf = *reinterpret_cast<const float4*>(kb + kg * 8);
            const float4 vf = *reinterpret_cast<const float4*>(vb + kg * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kg].x, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, dof[kg].x, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kg].y, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, dof[kg].y, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kg].z, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, dof[kg].z, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kg].w, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, dof[kg].w, dp, 0, 0, 0);
        }
        // dS^T = P^T o (dP^T - D)   (keys >= T contribute nothing)
        const int key0 = tile * TT + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2);
            const float p = key < T ? __expf(s[r] - Lq) : 0.f;
            s[r] = p * (dp[r] - Dq);
        }
        // dQ^T[c][q] += K^T[c][key] dS^T[key][q]
        const float* kt = kbuf + buf * TILE + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                dq[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[krow * KP + ct * 32], s[r], dq[ct], 0, 0, 0);
        }
        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    store_rows<CH, CT>(smem, dq, qscale, dqdst + (size_t)n * Tq * lddq + h * hsq, lddq, qb * BB, Tq, wave, lq, hi, tid);
}

template <int CH>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const float* __restrict__ qsrc, int ldq, int hsq,
                                                           const float* __restrict__ ksrc, const float* __restrict__ vsrc,
                                                           int ldkv, int hskv, const float* __restrict__ dout, int lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ Dv,
                                                           float* __restrict__ dkdst, float* __restrict__ dvdst, int lddkv,
                                                           int Tq, int T, int heads, float qscale, float scale) {
    constexpr int KP = CH + 4;
    constexpr int CT = (CH + 31) / 32;
    constexpr int KG = CH / 8;
    constexpr int KV4 = TT * CH / 4;
    constexpr int SLOTS = (KV4 + 255) / 256;
    constexpr int TILE = TT * KP;
    constexpr int STAGE = BB * (CH + 1);
    constexpr int SMEM = (4 * TILE + 4 * TT > STAGE ? 4 * TILE + 4 * TT : STAGE) + 64;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* qbuf = smem;                  // [2][TT][KP]  scaled queries
    float* dbuf = smem + 2 * TILE;       // [2][TT][KP]  dO
    float* lbuf = smem + 4 * TILE;       // [2][TT] lse, then [2][TT] D

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, lq = lane & 31;
    const int kblocks = (T + BB - 1) / BB;
    const int kb_ = blockIdx.x % kblocks, nh = blockIdx.x / kblocks, h = nh % heads, n = nh / heads;
    const float* qbase = qsrc + (size_t)n * Tq * ldq + h * hsq;
    const float* kbase = ksrc + (size_t)n * T * ldkv + h * hskv;
    const float* vbase = vsrc + (size_t)n * T * ldkv + h * hskv;
    const float* dob = dout + (size_t)n * Tq * lddo + h * CH;
    const float* lrow = lse + ((size_t)n * heads + h) * Tq;
    const float* drow = Dv + ((size_t)n * heads + h) * Tq;
    const int key = kb_ * BB + wave * 32 + lq;
    const bool kok = key < T;

    float4 kf[KG], vf[KG];               // K^T (scaled) and V^T as B operands: lane = key, k = channel
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        kf[kg] = vf[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kok) {
            const float4 a = *reinterpret_cast<const float4*>(kbase + (size_t)key * ldkv + kg * 8 + hi * 4);
            kf[kg] = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
            vf[kg] = *reinterpret_cast<const float4*>(vbase + (size_t)key * ldkv + kg * 8 + hi * 4);
        }
    }
    f32x16 dk[CT], dv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[ct][r] = dv[ct][r] = 0.f;

    float4 qreg[SLOTS], dreg[SLOTS];
    float lreg = 0.f, ddreg = 0.f;
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            const int qq = tile * TT + f / (CH / 4), c = (f % (CH / 4)) * 4;
            qreg[s] = dreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < KV4 && qq < Tq) {
                const float4 a = *reinterpret_cast<const float4*>(qbase + (size_t)qq * ldq + c);
                qreg[s] = make_float4(a.x * qscale, a.y * qscale, a.z * qscale, a.w * qscale);
                dreg[s] = *reinterpret_cast<const float4*>(dob + (size_t)qq * lddo + c);
            }
        }
        if (tid < TT) {
            const int qq = tile * TT + tid;
            lreg = qq < Tq ? lrow[qq] : 0.f;
            ddreg = qq < Tq ? drow[qq] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int f = tid + s * 256;
            if (f < KV4) {
                const int qq = f / (CH / 4), c = (f % (CH / 4)) * 4;
                *reinterpret_cast<float4*>(qbuf + buf * TILE + qq * KP + c) = qreg[s];
                *reinterpret_cast<float4*>(dbuf + buf * TILE + qq * KP + c) = dreg[s];
            }
        }
        if (tid < TT) {
            lbuf[buf * TT + tid] = lreg;
            lbuf[2 * TT + buf * TT + tid] = ddreg;
        }
    };
    for (int i = tid; i < SMEM; i += 256) smem[i] = 0.f;
    __syncthreads();
    const int ntiles = (Tq + TT - 1) / TT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);
        // S[q][key] = Qs K^T,  dP[q][key] = dO V^T      (rows = queries of the tile, lane = my key)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        const float* qb = qbuf + buf * TILE + lq * KP + hi * 4;
        const float* db = dbuf + buf * TILE + lq * KP + hi * 4;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const float4 a = *reinterpret_cast<const float4*>(qb + kg * 8);
            const float4 d = *reinterpret_cast<const float4*>(db + kg * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kf[kg].x, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, vf[kg].x, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kf[kg].y, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, vf[kg].y, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kf[kg].z, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(d.z, vf[kg].z, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kf[kg].w, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(d.w, vf[kg].w, dp, 0, 0, 0);
        }
        const int q0 = tile * TT + 4 * hi;
        f32x16 p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int qq = q0 + (r & 3) + 8 * (r >> 2);
            const float pr = (qq < Tq && kok) ? __expf(s[r] - lbuf[buf * TT + ql]) : 0.f;
            p[r] = pr;
            s[r] = pr * (dp[r] - lbuf[2 * TT + buf * TT + ql]);        // dS
        }
        // dV^T[c][key] += dO^T[c][q] P[q][key] ;  dK^T[c][key] += Qs^T[c][q] dS[q][key]
        const float* dt = dbuf + buf * TILE + lq;
        const float* qt = qbuf + buf * TILE + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dt[qrow * KP + ct * 32], p[r], dv[ct], 0, 0, 0);
                dk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(qt[qrow * KP + ct * 32], s[r], dk[ct], 0, 0, 0);
            }
        }
        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    const size_t doff = (size_t)n * T * lddkv + h * hskv;
    store_rows<CH, CT>(smem, dk, scale, dkdst + doff, lddkv, kb_ * BB, T, wave, lq, hi, tid);
    store_rows<CH, CT>(smem, dv, 1.0f, dvdst + doff, lddkv, kb_ * BB, T, wave, lq, hi, tid);
}

template <int CH>
void launch_bwd(const float* q, int ldq, int hsq, const float* k, const float* v, int ldkv, int hskv, const float* dout, int lddo,
                const float* lse, const float* D, float* dq, int lddq, float* dk, float* dv, int lddkv, int N, int Tq, int Tk,
                int heads, float qscale, float kscale, hipStream_t st) {
    const dim3 gq((unsigned)((long long)N * heads * ((Tq + BB - 1) / BB))), gk((unsigned)((long long)N * heads * ((Tk + BB - 1) / BB)));
    hipLaunchKernelGGL(attn_bwd_dq_kernel<CH>, gq, dim3(256), 0, st, q, ldq, hsq, k, v, ldkv, hskv, dout, lddo, lse, D, dq, lddq,
                       Tq, Tk, heads, qscale, kscale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<CH>, gk, dim3(256), 0, st, q, ldq, hsq, k, v, ldkv, hskv, dout, lddo, lse, D, dk, dv,
                       lddkv, Tq, Tk, heads, qscale, kscale);
}

int attention_bwd_common(const float* q, int ldq, int hsq, const float* k, const float* v, int ldkv, int hskv, const float* out,
                         int ldo, const float* dout, int lddo, const float* lse, float* dwork, float* dq, int lddq, float* dk,
                         float* dv, int lddkv, int N, int Tq, int Tk, int heads, int ch, float qscale, float kscale,
                         hipStream_t st) {
    const long long total = (long long)N * heads * Tq;
    int pb = (int)((total + 255) / 256);
    if (pb > 4096) pb = 4096;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(pb), dim3(256), 0, st, out, ldo, dout, lddo, dwork, N, Tq, heads, ch);
#define BBDM_ATTN_BWD(CH) \
    launch_bwd<CH>(q, ldq, hsq, k, v, ldkv, hskv, dout, lddo, lse, dwork, dq, lddq, dk, dv, lddkv, N, Tq, Tk, heads, qscale, kscale, st)
    if (ch == 64) BBDM_ATTN_BWD(64); else if (ch == 32) BBDM_ATTN_BWD(32); else BBDM_ATTN_BWD(16);
#undef BBDM_ATTN_BWD
    return 0;
}

}  // namespace

extern "C" int bbdm_attention_bwd_f32(const float* qkv, int ldq, const float* out, int ldo, const float* dout, int lddo,
                                      const float* lse, float* dwork, float* dqkv, int lddq, int N, int T, int heads, int ch,
                                      int new_order, void* stream) {
    BBDM_REQUIRE(qkv && out && dout && lse && dwork && dqkv, "attention_bwd: null pointer");
    BBDM_REQUIRE(N > 0 && T > 0 && heads > 0 && (ch == 16 || ch == 32 || ch == 64), "attention_bwd: bad shape (ch=%d)", ch);
    BBDM_REQUIRE(ldq % 4 == 0 && lddo % 4 == 0 && ldo % 4 == 0 && ldq >= 3 * heads * ch && lddq >= 3 * heads * ch &&
                     ldo >= heads * ch && lddo >= heads * ch,
                 "attention_bwd: bad pitch");
    BBDM_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout) & 15) == 0, "attention_bwd: 16-byte alignment");
    const float scale = 1.0f / sqrtf(sqrtf((float)ch));
    const int C = heads * ch;
    // QKVAttention (new order): q | k | v thirds, head h at h * ch inside each; QKVAttentionLegacy: per-head (q, k, v) triples
    const int hs = new_order ? ch : 3 * ch, ko = new_order ? C : ch, vo = new_order ? 2 * C : 2 * ch;
    attention_bwd_common(qkv, ldq, hs, qkv + ko, qkv + vo, ldq, hs, out, ldo, dout, lddo, lse, dwork, dqkv, lddq, dqkv + ko,
                         dqkv + vo, lddq, N, T, T, heads, ch, scale, scale, (hipStream_t)stream);
    BBDM_CHECK_LAUNCH("attention_bwd");
    return BBDM_OK;
}

// Backward of bbdm_cross_attention_f32 (CrossAttention.forward, attention.py:170-194): dq [N][Tq][heads*ch] (pitch lddq),
// dk / dv [N][Tk][heads*ch] (pitch lddkv), all overwritten.  lse: the forward's [N][heads][Tq] log-sum-exp; dwork: N*heads*Tq
// floats of scratch.
extern "C" int bbdm_cross_attention_bwd_f32(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out,
                                            int ldo, const float* dout, int lddo, const float* lse, float* dwork, float* dq,
                                            int lddq, float* dk, float* dv, int lddkv, int N, int Tq, int Tk, int heads, int ch,
                                            void* stream) {
    BBDM_REQUIRE(q && k && v && out && dout && lse && dwork && dq && dk && dv, "cross_attention_bwd: null pointer");
    BBDM_REQUIRE(N > 0 && Tq > 0 && Tk > 0 && heads > 0 && (ch == 16 || ch == 32 || ch == 64),
                 "cross_attention_bwd: bad shape (ch=%d)", ch);
    const int C = heads * ch;
    BBDM_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && lddo % 4 == 0 && ldo % 4 == 0 && ldq >= C && ldkv >= C && ldo >= C && lddo >= C &&
                     lddq >= C && lddkv >= C, "cross_attention_bwd: bad pitch");
    BBDM_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout) & 15) == 0,
                 "cross_attention_bwd: 16-byte alignment");
    attention_bwd_common(q, ldq, ch, k, v, ldkv, ch, out, ldo, dout, lddo, lse, dwork, dq, lddq, dk, dv, lddkv, N, Tq, Tk, heads,
                         ch, 1.0f / sqrtf((float)ch), 1.0f, (hipStream_t)stream);
    BBDM_CHECK_LAUNCH("cross_attention_bwd");
    return BBDM_OK;
}
