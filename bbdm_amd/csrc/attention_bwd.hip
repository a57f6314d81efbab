// attention_bwd.hip -- backward of the streaming fp32 attention (training path).
//
// Reference: autograd through QKVAttentionLegacy / QKVAttention (openaimodel.py:359-375, 398-413); the reference
// wraps the whole AttentionBlock in CheckpointFunction (util.py:119-148, openaimodel.py:318), i.e. it re-runs the
// forward in backward -- here the probabilities are recomputed from the saved per-query log-sum-exp instead.
//
//   qs = q s, ks = k s (s = ch^-1/4),  S = qs.ks,  P = exp(S - lse),  O = P V
//   D[t]  = sum_c dO[t,c] O[t,c]                                   (prep kernel)
//   dS    = P o (dO.V^T - D),  dq = s sum_s dS ks,  dk = s sum_t dS qs,  dv = sum_t P dO
// Two kernels in the same transposed-operand style as the forward (a query / a key is a LANE, so lse and D are
// per-lane scalars and the recomputed P / dS registers are used in place as the next MFMA's B operand):
//   attn_bwd_dq  : workgroup = 128 queries, streams key tiles   -> dq
//   attn_bwd_dkv : workgroup = 128 keys,    streams query tiles -> dk, dv
#include "common.h"

namespace {

constexpr int TT = 32;         // streamed tile (keys in dq, queries in dkv)
constexpr int BB = 128;        // rows owned by a block

__device__ __forceinline__ int part_off(int new_order, int heads, int CH, int h, int part) {
    return new_order ? part * heads * CH + h * CH : h * 3 * CH + part * CH;
}

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
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const float* __restrict__ qkv, int ldq,
                                                          const float* __restrict__ dout, int lddo,
                                                          const float* __restrict__ lse, const float* __restrict__ Dv,
                                                          float* __restrict__ dqkv, int lddq, int T, int heads,
                                                          int new_order, float scale) {
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
    const int qblocks = (T + BB - 1) / BB;
    const int qb = blockIdx.x % qblocks, nh = blockIdx.x / qblocks, h = nh % heads, n = nh / heads;
    const int qoff = part_off(new_order, heads, CH, h, 0), koff = part_off(new_order, heads, CH, h, 1),
              voff = part_off(new_order, heads, CH, h, 2);
    const float* base = qkv + (size_t)n * T * ldq;
    const float* dob = dout + (size_t)n * T * lddo + h * CH;
    const int q = qb * BB + wave * 32 + lq;
    const bool qok = q < T;

    float4 qf[KG], dof[KG];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        qf[kg] = dof[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qok) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)q * ldq + qoff + kg * 8 + hi * 4);
            qf[kg] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
            dof[kg] = *reinterpret_cast<const float4*>(dob + (size_t)q * lddo + kg * 8 + hi * 4);
        }
    }
    const float Lq = qok ? lse[((size_t)n * heads + h) * T + q] : 0.f;
    const float Dq = qok ? Dv[((size_t)n * heads + h) * T + q] : 0.f;

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
                const float* row = base + (size_t)key * ldq;
                const float4 kv = *reinterpret_cast<const float4*>(row + koff + c);
                kreg[s] = make_float4(kv.x * scale, kv.y * scale, kv.z * scale, kv.w * scale);
                vreg[s] = *reinterpret_cast<const float4*>(row + voff + c);
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
    store_rows<CH, CT>(smem, dq, scale, dqkv + (size_t)n * T * lddq + qoff, lddq, qb * BB, T, wave, lq, hi, tid);
}

template <int CH>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const float* __restrict__ qkv, int ldq,
                                                           const float* __restrict__ dout, int lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ Dv,
                                                           float* __restrict__ dqkv, int lddq, int T, int heads,
                                                           int new_order, float scale) {
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
    const int qoff = part_off(new_order, heads, CH, h, 0), koff = part_off(new_order, heads, CH, h, 1),
              voff = part_off(new_order, heads, CH, h, 2);
    const float* base = qkv + (size_t)n * T * ldq;
    const float* dob = dout + (size_t)n * T * lddo + h * CH;
    const float* lrow = lse + ((size_t)n * heads + h) * T;
    const float* drow = Dv + ((size_t)n * heads + h) * T;
    const int key = kb_ * BB + wave * 32 + lq;
    const bool kok = key < T;

    float4 kf[KG], vf[KG];               // K^T (scaled) and V^T as B operands: lane = key, k = channel
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        kf[kg] = vf[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kok) {
            const float* row = base + (size_t)key * ldq;
            const float4 a = *reinterpret_cast<const float4*>(row + koff + kg * 8 + hi * 4);
            kf[kg] = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
            vf[kg] = *reinterpret_cast<const float4*>(row + voff + kg * 8 + hi * 4);
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
            if (f < KV4 && qq < T) {
                const float4 a = *reinterpret_cast<const float4*>(base + (size_t)qq * ldq + qoff + c);
                qreg[s] = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
                dreg[s] = *reinterpret_cast<const float4*>(dob + (size_t)qq * lddo + c);
            }
        }
        if (tid < TT) {
            const int qq = tile * TT + tid;
            lreg = qq < T ? lrow[qq] : 0.f;
            ddreg = qq < T ? drow[qq] : 0.f;
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
    const int ntiles = (T + TT - 1) / TT;
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
            const float pr = (qq < T && kok) ? __expf(s[r] - lbuf[buf * TT + ql]) : 0.f;
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
    float* dbase = dqkv + (size_t)n * T * lddq;
    store_rows<CH, CT>(smem, dk, scale, dbase + koff, lddq, kb_ * BB, T, wave, lq, hi, tid);
    store_rows<CH, CT>(smem, dv, 1.0f, dbase + voff, lddq, kb_ * BB, T, wave, lq, hi, tid);
}

template <int CH>
void launch_bwd(const float* qkv, int ldq, const float* dout, int lddo, const float* lse, const float* D, float* dqkv,
                int lddq, int N, int T, int heads, int new_order, float scale, hipStream_t st) {
    const int blocks = (T + BB - 1) / BB;
    const dim3 grid((unsigned)((long long)N * heads * blocks));
    hipLaunchKernelGGL(attn_bwd_dq_kernel<CH>, grid, dim3(256), 0, st, qkv, ldq, dout, lddo, lse, D, dqkv, lddq, T, heads,
                       new_order, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<CH>, grid, dim3(256), 0, st, qkv, ldq, dout, lddo, lse, D, dqkv, lddq, T, heads,
                       new_order, scale);
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
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf(sqrtf((float)ch));
    const long long total = (long long)N * heads * T;
    int pb = (int)((total + 255) / 256);
    if (pb > 4096) pb = 4096;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(pb), dim3(256), 0, st, out, ldo, dout, lddo, dwork, N, T, heads, ch);
    if (ch == 64) launch_bwd<64>(qkv, ldq, dout, lddo, lse, dwork, dqkv, lddq, N, T, heads, new_order, scale, st);
    else if (ch == 32) launch_bwd<32>(qkv, ldq, dout, lddo, lse, dwork, dqkv, lddq, N, T, heads, new_order, scale, st);
    else launch_bwd<16>(qkv, ldq, dout, lddo, lse, dwork, dqkv, lddq, N, T, heads, new_order, scale, st);
    BBDM_CHECK_LAUNCH("attention_bwd");
    return BBDM_OK;
}
