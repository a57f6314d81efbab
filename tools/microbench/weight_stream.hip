// weight_stream.hip -- how fast can ONE launch stream a layer's pre-split weights (read exactly once) through the LDS of 256 CUs?
//
// The small-latent configurations (C5 / C1: a few hundred GEMM rows) are bound by their weights: the 16 tile GEMMs of a 4x4 x 1024 ->
// 1024 layer read 100 MB of U planes once and take 48 us = 2.1 TB/s, far below what HBM delivers, and deeper LDS rings in
// gemm_bf3p_pipe_kernel moved that number by nothing.  This program isolates the data movement of that kernel --
// same grid (256 workgroups of 4 waves), same unit layout ([row group][chunk][plane][1 KB]), same 1 KB LDS-DMA copies, same
// wait + barrier per chunk, optionally the 24 MFMAs per wave and chunk and the A operand re-read from L2 -- and varies one thing at a
// time: ring depth, barrier, A traffic, waves per workgroup, workgroups, and a plain coalesced register-load stream as the ceiling.
//     hipcc -O3 --offload-arch=gfx950 weight_stream.hip -o weight_stream && ./weight_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int UNIT = 1024;

struct Args {
    const unsigned char* A;   // [16 points][4 row groups][nfull][3][UNIT]  (re-read by the 8 column tiles of a point: L2)
    const unsigned char* B;   // [16 points][32 row groups][nfull][3][UNIT] (read once)
    int nfull, n;             // chunks of the full contraction, chunks one workgroup walks (split-K: nfull / n splits)
    int with_a, barrier, mfma;
    int hot;                  // every workgroup reads the SAME units (point 0, column tile 0): L2 hits -- what one CU can pull, not HBM
    unsigned* sink;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const unsigned char* unit, unsigned lane16, unsigned char* lds_wave_base) {
    const unsigned long long u = (unsigned long long)(uintptr_t)unit;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    const unsigned char* base = (const unsigned char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + lane16),
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// workgroup g: point p = g % 16, column tile ct = (g / 16) % 8, split z = g / 128 -- 4 B row groups (+ 4 A row groups of the point)
template <int NS, int NW, int RG, int WORK = 1>   // LDS stages, waves, row groups per operand (4: 128-wide tiles, 8: 256-wide), what a wave does per chunk
                                                 // when a.mfma is set: 1 = 24 MFMAs, 2 = s_sleep of about their duration, 3 = VALU FMAs of about their duration
__global__ void __launch_bounds__(NW * 64, 1) stream_lds(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NA = RG * 3, NU = 2 * NA, STAGE = NU * UNIT, KMAX = NU / NW;
    static_assert(NU % NW == 0, "");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane16 = lane * 16;
    const int g = blockIdx.x;
    const int ctiles = 32 / RG;
    const int p = a.hot ? 0 : g % 16, ct = a.hot ? 0 : (g / 16) % ctiles, z = g / (16 * ctiles);
    const size_t gstride = (size_t)a.nfull * 3 * UNIT;
    const unsigned char* src[KMAX];
    bool live[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int u = wave + k * NW, ub = u - NA;
        live[k] = u >= NA || a.with_a;
        src[k] = (u < NA ? a.A + ((size_t)(p * 4 + (u / 3) % 4)) * gstride + (u % 3) * UNIT
                         : a.B + ((size_t)(p * 32 + ct * RG + ub / 3)) * gstride + (ub % 3) * UNIT) +
                 (size_t)z * a.n * 3 * UNIT;
    }
    auto issue = [&](int chunk, unsigned char* st) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (live[k]) glds16(src[k] + (size_t)chunk * (3 * UNIT), lane16, st + (wave + k * NW) * UNIT);
    };
    const int n = a.n;
#pragma unroll
    for (int k = 0; k < NS - 1; ++k)
        if (k < n) issue(k, smem + k * STAGE);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 fa, fb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(float)(lane + j); fb[j] = (__bf16)(float)(lane - j); }
    i32x4 x = {0, 0, 0, 0};
    // iteration c: wait for chunk c (NS - 2 younger chunks may stay in flight), barrier, request chunk c + NS - 1 into the stage chunk
    // c - 1 was read from, consume chunk c
    for (int chunk = 0; chunk < n; ++chunk) {
        const int younger = min(n - 1 - chunk, NS - 2);
        if (!a.with_a) {           // (half the copies are dead: count only live ones -- KMAX / 2 per chunk when NW divides NA)
            if (younger >= 3) wait_vmcnt<3 * (KMAX / 2 > 0 ? KMAX / 2 : 1)>();
            else if (younger == 2) wait_vmcnt<2 * (KMAX / 2 > 0 ? KMAX / 2 : 1)>();
            else if (younger == 1) wait_vmcnt<1 * (KMAX / 2 > 0 ? KMAX / 2 : 1)>();
            else wait_vmcnt<0>();
        } else {
            if (younger >= 3) wait_vmcnt<3 * KMAX>();
            else if (younger == 2) wait_vmcnt<2 * KMAX>();
            else if (younger == 1) wait_vmcnt<1 * KMAX>();
            else wait_vmcnt<0>();
        }
        if (a.barrier) __builtin_amdgcn_s_barrier();
        if (chunk + NS - 1 < n) issue(chunk + NS - 1, smem + ((chunk + NS - 1) % NS) * STAGE);
        const i32x4 v = *(const i32x4*)(smem + (chunk % NS) * STAGE + ((wave * 3) % NU) * UNIT + lane16);
        x ^= v;
        if (WORK == 1) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
        } else if (WORK == 2) {
#pragma unroll
            for (int t = 0; t < 6; ++t) __builtin_amdgcn_s_sleep(2);          // 6 x 128 clocks
        } else if (WORK == 3) {
            for (int t = 0; t < 12; ++t)                                       // 12 x 16 fmas x 4 clocks = 768 clocks
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[0][j] = __builtin_fmaf(acc[0][j], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678 && s == 1.25f) a.sink[0] = 1;
}

// ... with the work SPLIT BY WAVE: waves 0 - 3 only request the copies and wait for them, waves 4 - 7 (the second wave of each SIMD) only
// read fragments and issue the 24 MFMAs per chunk -- does a wave's own MFMA stream hold up the LDS-DMA copies IT requested?
template <int NS, int RG, int WORK>
__global__ void __launch_bounds__(512, 1) stream_lds_spec(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NA = RG * 3, NU = 2 * NA, STAGE = NU * UNIT, KMAX = NU / 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave < 4;
    const int pw = wave & 3;
    const unsigned lane16 = lane * 16;
    const int g = blockIdx.x, ctiles = 32 / RG;
    const int p = a.hot ? 0 : g % 16, ct = a.hot ? 0 : (g / 16) % ctiles, z = g / (16 * ctiles);
    const size_t gstride = (size_t)a.nfull * 3 * UNIT;
    const unsigned char* src[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int u = pw + k * 4, ub = u - NA;
        src[k] = (u < NA ? a.A + ((size_t)(p * 4 + (u / 3) % 4)) * gstride + (u % 3) * UNIT
                         : a.B + ((size_t)(p * 32 + ct * RG + ub / 3)) * gstride + (ub % 3) * UNIT) +
                 (size_t)z * a.n * 3 * UNIT;
    }
    auto issue = [&](int chunk, unsigned char* st) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) glds16(src[k] + (size_t)chunk * (3 * UNIT), lane16, st + (pw + k * 4) * UNIT);
    };
    const int n = a.n;
    if (producer) {
#pragma unroll
        for (int k = 0; k < NS - 1; ++k)
            if (k < n) issue(k, smem + k * STAGE);
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 fa, fb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(float)(lane + j); fb[j] = (__bf16)(float)(lane - j); }
    i32x4 x = {0, 0, 0, 0};
    for (int chunk = 0; chunk < n; ++chunk) {
        if (producer) {
            const int younger = min(n - 1 - chunk, NS - 2);
            if (younger >= 3) wait_vmcnt<3 * KMAX>();
            else if (younger == 2) wait_vmcnt<2 * KMAX>();
            else if (younger == 1) wait_vmcnt<1 * KMAX>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (producer) {
            if (chunk + NS - 1 < n) issue(chunk + NS - 1, smem + ((chunk + NS - 1) % NS) * STAGE);
        } else {
            const i32x4 v = *(const i32x4*)(smem + (chunk % NS) * STAGE + ((pw * 3) % NU) * UNIT + lane16);
            x ^= v;
            if (WORK == 1) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[i][j];
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678 && sum == 1.25f) a.sink[0] = 1;
}

// the same movement through REGISTERS: a thread requests its 16-B pieces of the chunks c .. c + D - 1 with global_load_dwordx4,
// writes chunk c to LDS with ds_write_b128 when it arrives, barrier, one fragment read -- is the LDS-DMA path what limits a CU?
template <int D, int RG, int WORK = 0>
__global__ void __launch_bounds__(256, 1) stream_regs(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NU = 2 * RG * 3, STAGE = NU * UNIT, PER = STAGE / 16 / 256;      // 16-B pieces per thread and chunk (6 at RG = 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, ctiles = 32 / RG;
    const int p = a.hot ? 0 : g % 16, ct = a.hot ? 0 : (g / 16) % ctiles, z = g / (16 * ctiles);
    const size_t gstride = (size_t)a.nfull * 3 * UNIT;
    const unsigned char* src[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int u = wave + k * 4, ub = u - RG * 3;                            // unit of the stage, as in stream_lds
        src[k] = (u < RG * 3 ? a.A + ((size_t)(p * 4 + (u / 3) % 4)) * gstride + (u % 3) * UNIT
                             : a.B + ((size_t)(p * 32 + ct * RG + ub / 3)) * gstride + (ub % 3) * UNIT) +
                 (size_t)z * a.n * 3 * UNIT + lane * 16;
    }
    i32x4 r[D][PER];
    const int n = a.n;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 fa, fb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(float)(lane + j); fb[j] = (__bf16)(float)(lane - j); }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < PER; ++k) r[d][k] = d < n ? *(const i32x4*)(src[k] + (size_t)d * 3 * UNIT) : i32x4{0, 0, 0, 0};
    i32x4 x = {0, 0, 0, 0};
    for (int c0 = 0; c0 < n; c0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int chunk = c0 + d;
            if (chunk < n) {
                unsigned char* st = smem + (chunk & 1) * STAGE;
#pragma unroll
                for (int k = 0; k < PER; ++k) *(i32x4*)(st + (wave + k * 4) * UNIT + lane * 16) = r[d][k];
#pragma unroll
                for (int k = 0; k < PER; ++k)
                    if (chunk + D < n) r[d][k] = *(const i32x4*)(src[k] + (size_t)(chunk + D) * 3 * UNIT);
                __syncthreads();
                x ^= *(const i32x4*)(st + ((wave * 3) % NU) * UNIT + lane * 16);
                if (WORK == 1) {
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
                }
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[i][j];
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678 && sum == 1.25f) a.sink[0] = 1;
}

// the ceiling: every thread of a full-chip grid walks the buffer with coalesced 16-byte loads, DEPTH loads in flight per thread
template <int DEPTH>
__global__ void __launch_bounds__(256) stream_reg(const i32x4* __restrict__ b, size_t count, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    i32x4 x = {0, 0, 0, 0};
    for (; i + (DEPTH - 1) * stride < count; i += DEPTH * stride) {
        i32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(b + i + d * stride);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) x ^= v[d];
    }
    for (; i < count; i += stride) x ^= b[i];
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678) sink[0] = 1;
}

int main() {
    const int nfull = 64, NBUF = 8;                                   // K = 1024: 64 chunks; 8 weight sets so that no launch finds its
    const size_t bbytes = (size_t)16 * 32 * nfull * 3 * UNIT;         // weights in a cache (8 x 100 MB > the 256 MB Infinity Cache)
    const size_t abytes = (size_t)16 * 4 * nfull * 3 * UNIT;
    unsigned char *A, *B[NBUF];
    unsigned* sink;
    CK(hipMalloc(&A, abytes));
    CK(hipMemset(A, 1, abytes));
    for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&B[i], bbytes)); CK(hipMemset(B[i], i + 1, bbytes)); }
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("weights per launch: %.1f MB (read once), A operand %.1f MB (re-read by 8 column tiles); work: 0 none, 1 = 24 MFMAs per wave and chunk, 2 sleep, 3 valu\n", bbytes / 1e6, abytes / 1e6);

    auto time_it = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < NBUF; ++i) launch(i);
        CK(hipDeviceSynchronize());
        const int reps = 5 * NBUF;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(i % NBUF);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-72s %7.1f us  %5.2f TB/s of weights\n", name, us, bytes / us / 1e6);
    };
    int hot_mode = 0;
#define REG_CASE(D, RG, splits) REG_CASE_W(D, RG, splits, 0)
#define REG_CASE_W(D, RG, splits, mf)                                                                                                     \
    do {                                                                                                                            \
        const size_t lds = (size_t)2 * 2 * RG * 3 * UNIT;                                                                           \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_regs<D, RG, mf>), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                               (int)lds));                                                                                          \
        char nm[160];                                                                                                               \
        snprintf(nm, sizeof nm, "%sregister-staged, %d chunks in flight, %3d-col tiles, %d split(s) = %4d wgs, work %d", hot_mode ? "HOT " : "", \
                 D, RG * 32, splits, 16 * (32 / RG) * splits, mf);                                                                      \
        time_it(nm, [&](int i) {                                                                                                    \
            Args a{A, B[i], nfull, nfull / splits, 1, 1, 0, hot_mode, sink};                                                        \
            hipLaunchKernelGGL((stream_regs<D, RG, mf>), dim3(16 * (32 / RG) * splits), dim3(256), lds, 0, a);                          \
        }, (double)bbytes);                                                                                                         \
    } while (0)
#define SPEC_CASE(NS, RG, splits, mf)                                                                                               \
    do {                                                                                                                            \
        const size_t lds = (size_t)NS * 2 * RG * 3 * UNIT;                                                                          \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_lds_spec<NS, RG, mf>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)lds));                                                                                          \
        char nm[160];                                                                                                               \
        snprintf(nm, sizeof nm, "%sLDS-DMA ring %d, 4 copy waves + 4 MFMA waves, %3d-col tiles, %d split(s) = %4d wgs, work %d",    \
                 hot_mode ? "HOT " : "", NS, RG * 32, splits, 16 * (32 / RG) * splits, mf);                                         \
        time_it(nm, [&](int i) {                                                                                                    \
            Args a{A, B[i], nfull, nfull / splits, 1, 1, mf, hot_mode, sink};                                                       \
            hipLaunchKernelGGL((stream_lds_spec<NS, RG, mf>), dim3(16 * (32 / RG) * splits), dim3(512), lds, 0, a);                 \
        }, (double)bbytes);                                                                                                         \
    } while (0)
#define LDS_CASE(NS, NW, RG, splits, wa, bar, mf) LDS_CASE_W(NS, NW, RG, splits, wa, bar, mf)
#define LDS_CASE_W(NS, NW, RG, splits, wa, bar, mf)                                                                                   \
    do {                                                                                                                            \
        const size_t lds = (size_t)NS * 2 * RG * 3 * UNIT;                                                                          \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_lds<NS, NW, RG, mf>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                               (int)lds));                                                                                          \
        char nm[160];                                                                                                               \
        snprintf(nm, sizeof nm, "%sLDS-DMA ring %d, %2d waves, %3d-col tiles, %d split(s) = %4d wgs, A %d barrier %d work %d", hot_mode ? "HOT " : "", NS, NW, \
                 RG * 32, splits, 16 * (32 / RG) * splits, wa, bar, mf);                                                            \
        time_it(nm, [&](int i) {                                                                                                    \
            Args a{A, B[i], nfull, nfull / splits, wa, bar, mf, hot_mode, sink};                                                              \
            hipLaunchKernelGGL((stream_lds<NS, NW, RG, mf>), dim3(16 * (32 / RG) * splits), dim3(NW * 64), lds, 0, a);                  \
        }, (double)bbytes);                                                                                                         \
    } while (0)
    // the kernel as it is (2 stages, 4 waves, 128-col tiles, 2 splits = 256 workgroups)
    LDS_CASE(2, 4, 4, 2, 1, 1, 1);
    LDS_CASE(3, 4, 4, 2, 1, 1, 1);
    LDS_CASE(5, 4, 4, 2, 1, 1, 1);
    // ... the wave asleep / on the VALU instead of the matrix pipe for the same time
    LDS_CASE(2, 4, 4, 2, 1, 1, 2);
    LDS_CASE(5, 4, 4, 2, 1, 1, 2);
    LDS_CASE(2, 4, 4, 2, 1, 1, 3);
    LDS_CASE(5, 4, 4, 2, 1, 1, 3);
    // ... without MFMAs, without the barrier, without the A re-reads
    LDS_CASE(5, 4, 4, 2, 1, 1, 0);
    LDS_CASE(5, 4, 4, 2, 1, 0, 0);
    LDS_CASE(5, 4, 4, 2, 0, 1, 0);
    LDS_CASE(5, 4, 4, 2, 0, 0, 0);
    LDS_CASE(2, 4, 4, 2, 0, 0, 0);
    // more workgroups (deeper split), more waves
    LDS_CASE(2, 4, 4, 4, 1, 1, 1);
    LDS_CASE(3, 4, 4, 4, 1, 1, 1);
    LDS_CASE(3, 4, 4, 8, 1, 1, 1);
    LDS_CASE(3, 8, 4, 2, 1, 1, 1);
    LDS_CASE(5, 8, 4, 2, 1, 1, 1);
    LDS_CASE(3, 12, 4, 2, 1, 1, 0);
    // 256-column tiles: half the A traffic per weight byte
    LDS_CASE(3, 8, 8, 4, 1, 1, 1);
    LDS_CASE(3, 8, 8, 2, 1, 1, 1);
    LDS_CASE(3, 16, 8, 4, 1, 1, 1);
    // the same movement staged through registers; then everything again on L2-resident data (every workgroup reads the same units):
    // what ONE CU can pull per chunk, by LDS-DMA and through registers
    REG_CASE(2, 4, 2);
    REG_CASE(3, 4, 2);
    REG_CASE(4, 4, 2);
    hot_mode = 1;
    LDS_CASE(3, 4, 4, 2, 1, 1, 0);
    LDS_CASE(5, 4, 4, 2, 1, 1, 0);
    LDS_CASE(5, 4, 4, 2, 1, 1, 1);
    REG_CASE(2, 4, 2);
    REG_CASE(3, 4, 2);
    REG_CASE(4, 4, 2);
    SPEC_CASE(3, 4, 2, 0);
    SPEC_CASE(3, 4, 2, 1);
    SPEC_CASE(5, 4, 2, 1);
    REG_CASE_W(3, 4, 2, 1);
    REG_CASE_W(4, 4, 2, 1);
    hot_mode = 0;
    REG_CASE_W(3, 4, 2, 1);
    REG_CASE_W(4, 4, 2, 1);
    SPEC_CASE(3, 4, 2, 0);
    SPEC_CASE(3, 4, 2, 1);
    SPEC_CASE(5, 4, 2, 1);
    // the ceiling: coalesced register loads, full chip
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        char nm[160];
        snprintf(nm, sizeof nm, "register loads, %4d wgs x 256 threads, 4 x 16 B in flight per thread", wgs);
        time_it(nm, [&](int i) { hipLaunchKernelGGL((stream_reg<4>), dim3(wgs), dim3(256), 0, 0, (const i32x4*)B[i], bbytes / 16, sink); },
                (double)bbytes);
        snprintf(nm, sizeof nm, "register loads, %4d wgs x 256 threads, 8 x 16 B in flight per thread", wgs);
        time_it(nm, [&](int i) { hipLaunchKernelGGL((stream_reg<8>), dim3(wgs), dim3(256), 0, 0, (const i32x4*)B[i], bbytes / 16, sink); },
                (double)bbytes);
    }
    // an empty launch, for the fixed cost inside the numbers above
    time_it("empty kernel (launch-to-launch interval)", [&](int) { hipLaunchKernelGGL((stream_reg<4>), dim3(256), dim3(256), 0, 0, (const i32x4*)B[0], (size_t)0, sink); }, 0.0);
    return 0;
}
