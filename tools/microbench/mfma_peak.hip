// mfma_peak.hip -- what does an MI355X SUSTAIN on back-to-back v_mfma_f32_32x32x16_bf16, and at which clock?
//
// The dominant kernel of the C2 step (csrc/gemm_bf3p.hip) is priced against the datasheet's 2500 TFLOP/s dense bf16 peak.  DVFS
// (MI355X_MICROARCH.md "DVFS give-back") clocks the part to its power budget, so a stream of MFMAs on REAL operands does not run
// at the 2.4 GHz the datasheet figure assumes.  This program measures that ceiling directly: every SIMD of every CU holds WAVES
// waves that issue nothing but MFMAs (4 independent accumulators each, operands resident in registers), for ~SECONDS seconds,
//   (a) on all-zero operands (no toggling in the multiplier arrays: the clock-limited rate),
//   (b) on random bf16 operands drawn like the step's data (N(0,1) values, a different fragment pair per MFMA),
//   (c) on random operands with the three-plane magnitudes of the bf16x3 split (plane k scaled by 2^-8k),
// and prints TFLOP/s, the shader clock the kernel saw (s_memtime shader cycles per wall_clock64 100 MHz tick) and the MFMA issue
// interval in shader cycles per SIMD.  `tools/run_mfma_peak.sh` samples rocm-smi power / clocks beside it.
//     hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak [seconds] [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NFRAG = 4;            // distinct A and B fragments per wave, rotated so that consecutive MFMAs see different operands

// One wave: `iters` rounds of 4 x NACC MFMAs (NACC accumulators x 4 fragment pairs), nothing else in the loop.
// clocks[0..1]: shader cycles / 100 MHz ticks of block 0's loop; clocks[2] / clocks[3]: earliest loop start / latest loop end over ALL
// waves (atomicMin / atomicMax on the chip-wide 100 MHz counter): the burst as the chip saw it, launch overheads excluded.
template <int NACC>
__global__ void __launch_bounds__(256) mfma_stream_t(const bf16x8* __restrict__ frags, float* __restrict__ sink, long iters,
                                                     unsigned long long* __restrict__ clocks) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63;      // 64 fragment sets in memory
    bf16x8 a[NFRAG], b[NFRAG];
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) {
        a[i] = frags[((wave * 2 + 0) * NFRAG + i) * 64 + lane];
        b[i] = frags[((wave * 2 + 1) * NFRAG + i) * 64 + lane];
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(f + i) % NFRAG], b[f], acc[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123456.789f) sink[0] = s;                      // keep the accumulators live
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
    if ((threadIdx.x & 63) == 0) { atomicMin(&clocks[2], w0); atomicMax(&clocks[3], w1); }
}
#define mfma_stream mfma_stream_t<4>

// The same stream WITH the operand traffic of csrc/gemm_bf3p.hip's main loop: per 24 MFMAs a wave re-reads its 12 fragments from LDS
// (ds_read_b128 at lane * 16) and the workgroup's 16 waves together deposit 48 KB by LDS-DMA (global_load_lds_dwordx4, 3 per wave) from
// an L2-resident buffer.  TRAFFIC: 0 = none (registers only), 1 = the LDS reads, 2 = LDS reads + LDS-DMA copies, 3 = as 2 but one
// copy in three streams from a 2 GiB buffer (HBM: 16 KB of 48 KB per iteration, the tile GEMM's V share) and every wave stores
// 256 B per iteration to a streaming destination (the M tile's share) -- the HBM traffic of the real kernel, ~1.5 - 2 TB/s.
template <int TRAFFIC>
__global__ void __launch_bounds__(1024) mfma_traffic_stream(const bf16x8* __restrict__ frags, const unsigned char* __restrict__ src,
                                                            float* __restrict__ sink, long iters, unsigned long long* __restrict__ clocks,
                                                            float* __restrict__ stream_dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];            // 96 KB: two 48 KB stages
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 1024) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(frags)[i & 8191];
    __syncthreads();
    bf16x8 f[8];                     // (8 resident fragments: with 12 the 128-register budget of a 16-wave workgroup spilled)
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const bf16x8*>(lds + (wave * 12 + i) * 256 % (48 * 1024) + lane * 16);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (long it = 0; it < iters; ++it) {
        const int stage = (int)(it & 1) * 48 * 1024;
        if (TRAFFIC >= 2) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {           // 3 KB per wave and iteration: 16 waves x 3 = the 48 KB stage
                const unsigned char* g = (TRAFFIC >= 3 && q == 0)
                    ? src + ((((size_t)it * gridDim.x + blockIdx.x) * 16 + wave) & ((1u << 21) - 1)) * 1024 + lane * 16      // 2 GiB, streamed once
                    : src + (size_t)((blockIdx.x * 48 + wave * 3 + q) & 4095) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds + (stage ^ (48 * 1024)) + (wave * 3 + q) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {               // six term groups of four MFMAs, two fragment reloads behind each
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(2 * t + (i >> 1)) % 8], f[(2 * t + 4 + (i & 1)) % 8], acc[i], 0, 0, 0);
            if (TRAFFIC >= 1) {
                f[(2 * t) % 8] = *reinterpret_cast<const bf16x8*>(lds + stage + ((wave * 12 + 2 * t) * 1024) % (48 * 1024) + lane * 16);
                f[(2 * t + 1) % 8] = *reinterpret_cast<const bf16x8*>(lds + stage + ((wave * 12 + 2 * t + 1) * 1024) % (48 * 1024) + lane * 16);
            }
        }
        if (TRAFFIC >= 3)
            __builtin_nontemporal_store((float)lane, stream_dst + (((((size_t)it * gridDim.x + blockIdx.x) * 16 + wave) * 64 + lane) & ((1u << 29) - 1)));
        // (3: the HBM copy and the store are never consumed -- up to four iterations of them stay in flight, as the real kernel's
        // two-chunks-ahead pipeline keeps its copies; the L2-resident copies behind them are then not waited for either: this row
        // measures what the traffic does to the MFMA rate, not a correct pipeline)
        if (TRAFFIC == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (TRAFFIC >= 3) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); __syncthreads(); }
    }
    if (TRAFFIC >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123456.789f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
    if (lane == 0) { atomicMin(&clocks[2], w0); atomicMax(&clocks[3], w1); }
}

// a float4 copy at full HBM rate (the "transform" phase of the step between two GEMM bursts)
__global__ void __launch_bounds__(256) copy_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static uint16_t f2bf(float x) {              // round-to-nearest-even bf16
    uint32_t u; memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float gauss() {
    const float u1 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

static int burst_main(int argc, char** argv);

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "burst") == 0) return burst_main(argc - 1, argv + 1);
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int waves_per_simd = argc > 2 ? atoi(argv[2]) : 3;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clockRate %.0f MHz  waves per SIMD %d\n", prop.gcnArchName, cus, prop.clockRate / 1e3, waves_per_simd);
    const size_t nfr = 64 * 2 * NFRAG * 64;                 // fragments of 8 bf16
    std::vector<uint16_t> h(nfr * 8);
    bf16x8* d; float* sink; unsigned long long* clocks;
    CK(hipMalloc(&d, nfr * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clocks, 32));
    const int blocks = cus * waves_per_simd;                // 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    const double flop_per_wave_iter = 16.0 * 2.0 * 32 * 32 * 16;
    const char* names[3] = {"zero operands", "random N(0,1) bf16 operands", "random operands, bf16x3 plane magnitudes (1, 2^-8, 2^-16)"};
    for (int mode = 0; mode < 3; ++mode) {
        srand(1234);
        for (size_t i = 0; i < nfr; ++i) {
            const float scale = mode == 2 ? ldexpf(1.f, -8 * (int)((i / 64) % 3)) : 1.f;
            for (int k = 0; k < 8; ++k) h[i * 8 + k] = mode == 0 ? 0 : f2bf(gauss() * scale);
        }
        CK(hipMemcpy(d, h.data(), nfr * 16, hipMemcpyHostToDevice));
        // calibrate: iterations for ~0.25 s launches, then repeat launches for `seconds`
        long iters = 20000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_stream, dim3(blocks), dim3(256), 0, 0, d, sink, iters, clocks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 0) iters = (long)(iters * 250.0 / ms);
        }
        const int launches = (int)fmax(1.0, seconds / 0.25);
        double best = 0, sum = 0, clk_sum = 0, last_ms = 0;
        for (int l = 0; l < launches; ++l) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_stream, dim3(blocks), dim3(256), 0, 0, d, sink, iters, clocks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hc[2];
            CK(hipMemcpy(hc, clocks, 16, hipMemcpyDeviceToHost));
            const double tf = flop_per_wave_iter * iters * blocks * 4 / (ms * 1e-3) / 1e12;
            const double ghz = (double)hc[0] / ((double)hc[1] / 100e6) / 1e9;        // shader cycles per second of the 100 MHz wall clock
            best = fmax(best, tf); sum += tf; clk_sum += ghz; last_ms = ms;
        }
        const double tf = sum / launches, ghz = clk_sum / launches;
        // issue interval: one SIMD issues waves_per_simd x 16 x iters MFMAs per launch
        const double cyc_per_mfma = ghz * 1e9 * (last_ms * 1e-3) / ((double)waves_per_simd * 16.0 * iters);
        printf("%-62s %7.1f TFLOP/s mean (%7.1f best) over %d launches of %.0f ms  shader clock %.3f GHz  %.1f cycles per MFMA and SIMD  "
               "= %.3f of 2500\n", names[mode], tf, best, launches, last_ms, ghz, cyc_per_mfma, tf / 2500.0);
    }
    return 0;
}


// ---- burst mode (round 5):  ./mfma_peak burst [seconds]
// The continuous stream above is one duty cycle; the C2 step is another: 0.5 - 4.5 ms of tile GEMM between HBM-bound transforms that
// draw a third of the power.  Rows:
//   (i)   back-to-back MFMA launches of 0.5 / 1.5 / 5 / 50 ms (random operands) for `seconds`;
//   (ii)  ALTERNATING: a ~1.5 ms MFMA burst, then a ~0.5 ms float4 copy at full HBM rate, for `seconds`; TFLOP/s INSIDE the bursts from
//         the chip-wide 100 MHz counter (earliest loop start to latest loop end over all waves), first / middle / last thirds separately;
//   (iii) the same alternating schedule at 1, 2, 3 waves per SIMD and with 8 accumulators per wave.
// Shader clock of a burst = block 0's s_memtime cycles / its 100 MHz ticks.
template <int NACC>
static void launch_mfma(int blocks, const bf16x8* d, float* sink, long iters, unsigned long long* clk) {
    hipLaunchKernelGGL(mfma_stream_t<NACC>, dim3(blocks), dim3(256), 0, 0, d, sink, iters, clk);
}
struct BurstStat { double tf_mean, tf_min, tf_max, ghz, tf_third[3], burst_ms, gap_ms; int n; long iters; };

// traffic < 0: the register-only stream (wps 256-thread blocks per CU); traffic 0 / 1 / 2: mfma_traffic_stream<traffic>, ONE 16-wave
// workgroup per CU (4 waves per SIMD, 24 MFMAs per wave and iteration) -- the geometry of the 256 x 256 tile GEMM
static BurstStat run_schedule(int cus, int wps, int nacc, const bf16x8* d, float* sink, unsigned long long* clk_dev, int max_bursts,
                              double burst_ms, double copy_ms, double seconds, float4* csrc, float4* cdst, size_t copy_elems_per_ms,
                              int traffic = -1) {
    const int blocks = traffic >= 0 ? cus * 4 : cus * wps;                       // (x 4 waves below: 16 waves per CU in traffic mode)
    const double flop_iter = traffic >= 0 ? 24.0 * 2.0 * 32 * 32 * 16 : (double)nacc * 4.0 * 2.0 * 32 * 32 * 16;           // per wave and loop iteration
    auto go = [&](long iters, unsigned long long* c) {
        if (traffic >= 0) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(csrc);          // first 4 MB: L2 / MALL resident
            float* sd = reinterpret_cast<float*>(cdst);
            if (traffic == 0) hipLaunchKernelGGL(mfma_traffic_stream<0>, dim3(cus), dim3(1024), 96 * 1024, 0, d, src, sink, iters, c, sd);
            else if (traffic == 1) hipLaunchKernelGGL(mfma_traffic_stream<1>, dim3(cus), dim3(1024), 96 * 1024, 0, d, src, sink, iters, c, sd);
            else if (traffic == 2) hipLaunchKernelGGL(mfma_traffic_stream<2>, dim3(cus), dim3(1024), 96 * 1024, 0, d, src, sink, iters, c, sd);
            else hipLaunchKernelGGL(mfma_traffic_stream<3>, dim3(cus), dim3(1024), 96 * 1024, 0, d, src, sink, iters, c, sd);
        } else if (nacc == 8) launch_mfma<8>(blocks, d, sink, iters, c);
        else launch_mfma<4>(blocks, d, sink, iters, c);
    };
    // calibrate iterations for the burst length on a warm chip (a few launches, event-timed)
    long iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) go(iters, clk_dev);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        iters = (long)fmax(16.0, iters * burst_ms / (ms / 4));
    }
    const size_t copy_total = (size_t)(copy_elems_per_ms * copy_ms);       // elements; launched in pieces of at most the buffer
    const size_t copy_cap = (size_t)1 << 27;                               // 2 GiB / 16 B
    int n = (int)fmin((double)max_bursts, seconds * 1e3 / (burst_ms + copy_ms));
    if (n < 3) n = 3;
    std::vector<unsigned long long> init((size_t)n * 4);
    for (int l = 0; l < n; ++l) { init[l * 4] = 0; init[l * 4 + 1] = 0; init[l * 4 + 2] = ~0ull; init[l * 4 + 3] = 0; }
    CK(hipMemcpy(clk_dev, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    for (int l = 0; l < n; ++l) {                       // everything queued up front: no host round trip between a burst and its copy
        go(iters, clk_dev + (size_t)l * 4);
        for (size_t done = 0; done < copy_total; done += copy_cap)
            hipLaunchKernelGGL(copy_stream, dim3(cus * 8), dim3(256), 0, 0, csrc, cdst, copy_total - done < copy_cap ? copy_total - done : copy_cap);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)n * 4);
    CK(hipMemcpy(h.data(), clk_dev, h.size() * 8, hipMemcpyDeviceToHost));
    BurstStat st{};
    st.tf_min = 1e30; st.n = n; st.iters = iters;
    double third_sum[3] = {0, 0, 0}; int third_n[3] = {0, 0, 0};
    for (int l = 0; l < n; ++l) {
        const double sec = (double)(h[l * 4 + 3] - h[l * 4 + 2]) / 100e6;
        const double tf = flop_iter * iters * blocks * 4 / sec / 1e12;
        st.tf_mean += tf / n; st.tf_min = fmin(st.tf_min, tf); st.tf_max = fmax(st.tf_max, tf);
        st.ghz += (double)h[l * 4] / ((double)h[l * 4 + 1] / 100e6) / 1e9 / n;
        st.burst_ms += sec * 1e3 / n;
        if (l + 1 < n) st.gap_ms += (double)(h[(l + 1) * 4 + 2] - h[l * 4 + 3]) / 100e6 * 1e3 / (n - 1);
        const int t = l * 3 / n;
        third_sum[t] += tf; third_n[t]++;
    }
    for (int t = 0; t < 3; ++t) st.tf_third[t] = third_n[t] ? third_sum[t] / third_n[t] : 0;
    return st;
}

static int burst_main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("burst mode: device %s  CUs %d  %.1f s per row; random N(0,1) bf16 operands unless stated\n", prop.gcnArchName, cus, seconds);
    const size_t nfr = 64 * 2 * NFRAG * 64;
    std::vector<uint16_t> h(nfr * 8);
    bf16x8* d; float* sink; unsigned long long* clk;
    const int max_bursts = 8192;
    CK(hipMalloc(&d, nfr * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clk, (size_t)max_bursts * 32));
    const size_t cbytes = 2ull << 30;                           // 2 GiB source + 2 GiB destination: far beyond the 256 MB Infinity Cache
    float4 *csrc, *cdst;
    CK(hipMalloc(&csrc, cbytes)); CK(hipMalloc(&cdst, cbytes));
    CK(hipMemset(csrc, 1, cbytes));
    // copy rate: elements per ms (event-timed on a warm chip)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t call_n = cbytes / 16;
    double copy_ms = 0;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(copy_stream, dim3(cus * 8), dim3(256), 0, 0, csrc, cdst, call_n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        copy_ms = ms;
    }
    const size_t elems_per_ms = (size_t)(call_n / copy_ms);
    printf("float4 copy: %.2f TB/s (read + write) -> %.0f MB copied per ms\n", 2.0 * cbytes / (copy_ms * 1e-3) / 1e12, elems_per_ms * 16 / 1e6);
    auto fill = [&](int mode) {
        srand(1234);
        for (size_t i = 0; i < nfr; ++i)
            for (int k = 0; k < 8; ++k) h[i * 8 + k] = mode == 0 ? 0 : f2bf(gauss());
        CK(hipMemcpy(d, h.data(), nfr * 16, hipMemcpyHostToDevice));
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_traffic_stream<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_traffic_stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_traffic_stream<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_traffic_stream<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    auto row = [&](const char* name, int wps, int nacc, double bms, double cms, int traffic = -1) {
        const BurstStat s = run_schedule(cus, wps, nacc, d, sink, clk, max_bursts, bms, cms, seconds, csrc, cdst, elems_per_ms, traffic);
        if (traffic >= 0) { wps = 4; nacc = 6; }                                                            // 4 waves per SIMD, 24 MFMAs per iteration
        const double cyc = s.ghz * 1e9 * (s.burst_ms * 1e-3) / ((double)wps * nacc * 4.0 * s.iters);      // one SIMD issues wps x 4 nacc x iters MFMAs per burst
        printf("%-44s w/SIMD %d acc %d | %5d bursts of %6.3f ms, gap %6.3f ms | in-burst TFLOP/s mean %7.1f (min %7.1f max %7.1f; thirds %7.1f %7.1f %7.1f) "
               "| %.3f GHz, %.1f cycles per MFMA and SIMD | %.3f of 2500\n", name, wps, nacc, s.n, s.burst_ms, s.gap_ms, s.tf_mean, s.tf_min, s.tf_max,
               s.tf_third[0], s.tf_third[1], s.tf_third[2], s.ghz, cyc, s.tf_mean / 2500.0);
        fflush(stdout);
    };
    fill(1);
    // (i) back-to-back launches of one length
    for (double bms : {0.5, 1.5, 5.0, 50.0}) row("(i) back-to-back MFMA launches", 2, 4, bms, 0.0);
    // (ii) the step's duty cycle
    row("(ii) 1.5 ms MFMA / 0.5 ms copy", 2, 4, 1.5, 0.5);
    row("(ii) 1.5 ms MFMA / 1.5 ms copy", 2, 4, 1.5, 1.5);
    row("(ii) 0.5 ms MFMA / 0.5 ms copy", 2, 4, 0.5, 0.5);
    row("(ii) 4.5 ms MFMA / 1.5 ms copy", 2, 4, 4.5, 1.5);
    // (iii) occupancy / accumulator rows on the 1.5 / 0.5 schedule
    row("(iii) 1.5 / 0.5", 1, 4, 1.5, 0.5);
    row("(iii) 1.5 / 0.5", 1, 8, 1.5, 0.5);
    row("(iii) 1.5 / 0.5", 2, 8, 1.5, 0.5);
    row("(iii) 1.5 / 0.5", 3, 4, 1.5, 0.5);
    row("(iii) 1.5 / 0.5", 4, 4, 1.5, 0.5);
    // (iv) what the tile GEMM's operand traffic costs: the same MFMA count with its LDS fragment reads / LDS-DMA copies, 16 waves per CU
    row("(iv) GEMM geometry, registers only, 1.5 / 0.5", 4, 6, 1.5, 0.5, 0);
    row("(iv) + 12 ds_read_b128 per 24 MFMAs, 1.5 / 0.5", 4, 6, 1.5, 0.5, 1);
    row("(iv) + reads + 48 KB LDS-DMA per iter, 1.5 / 0.5", 4, 6, 1.5, 0.5, 2);
    row("(iv) + reads + LDS-DMA, back to back 1.5 ms", 4, 6, 1.5, 0.0, 2);
    row("(iv) + reads + DMA (1/3 from HBM) + stores, 1.5 / 0.5", 4, 6, 1.5, 0.5, 3);
    row("(iv) + reads + DMA (1/3 from HBM) + stores, back to back", 4, 6, 1.5, 0.0, 3);
    fill(0);
    row("zero operands: (i) 1.5 ms back to back", 2, 4, 1.5, 0.0);
    row("zero operands: (ii) 1.5 / 0.5", 2, 4, 1.5, 0.5);
    row("zero operands: (iii) 1 wave, 8 acc, 1.5 / 0.5", 1, 8, 1.5, 0.5);
    return 0;
}
