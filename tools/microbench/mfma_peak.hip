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

// One wave: `iters` rounds of 16 MFMAs (4 accumulators x 4 fragment pairs), nothing else in the loop.
__global__ void __launch_bounds__(256) mfma_stream(const bf16x8* __restrict__ frags, float* __restrict__ sink, long iters,
                                                   unsigned long long* __restrict__ clocks) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63;      // 64 fragment sets in memory
    bf16x8 a[NFRAG], b[NFRAG];
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) {
        a[i] = frags[((wave * 2 + 0) * NFRAG + i) * 64 + lane];
        b[i] = frags[((wave * 2 + 1) * NFRAG + i) * 64 + lane];
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(f + i) % NFRAG], b[f], acc[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123456.789f) sink[0] = s;                      // keep the accumulators live
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
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

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int waves_per_simd = argc > 2 ? atoi(argv[2]) : 3;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clockRate %.0f MHz  waves per SIMD %d\n", prop.gcnArchName, cus, prop.clockRate / 1e3, waves_per_simd);
    const size_t nfr = 64 * 2 * NFRAG * 64;                 // fragments of 8 bf16
    std::vector<uint16_t> h(nfr * 8);
    bf16x8* d; float* sink; unsigned long long* clocks;
    CK(hipMalloc(&d, nfr * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clocks, 16));
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
