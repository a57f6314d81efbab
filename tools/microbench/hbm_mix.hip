// hbm_mix.hip -- what HBM bandwidth does MI355X give a streaming kernel as a function of its read : write mix and store width?
// (The Winograd input transform writes 73 % of its bytes; is ~5 TB/s its floor?)   hipcc -O3 --offload-arch=gfx950 hbm_mix.hip -o hbm_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each thread: R reads of 16 B and Wn writes of 16 B per iteration, grid-stride over `n16` 16-byte elements
template <int R, int Wn>
__global__ void __launch_bounds__(256) mix16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 acc = make_uint4(1, 2, 3, 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 v = src[i + (size_t)r * n16];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
#pragma unroll
        for (int w = 0; w < Wn; ++w) {
            uint4 o = acc; o.x += w;
            dst[i + (size_t)w * n16] = o;
        }
        if (Wn == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) dst[0] = acc;     // keep the loads
    }
}
// 4-byte stores: a wave writes 256 B per instruction, Wn streams n4 elements apart (the input transform's store shape)
template <int R, int Wn>
__global__ void __launch_bounds__(256) mix4(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        unsigned acc = 1;
#pragma unroll
        for (int r = 0; r < R; ++r) acc += src[i + (size_t)r * n4];
#pragma unroll
        for (int w = 0; w < Wn; ++w) dst[i + (size_t)w * n4] = acc + w;
    }
}

template <class F>
static double time_ms(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t stream_bytes = 512ull << 20;           // per stream
    const int maxs = 6;
    void *src, *dst;
    CK(hipMalloc(&src, stream_bytes * maxs));
    CK(hipMalloc(&dst, stream_bytes * maxs));
    CK(hipMemset(src, 1, stream_bytes * maxs));
    CK(hipMemset(dst, 0, stream_bytes * maxs));
    const size_t n16 = stream_bytes / 16, n4 = stream_bytes / 4;
    const int reps = 5;
    printf("%-34s %8s %9s\n", "kernel (streams of 512 MiB)", "ms", "TB/s");
#define RUN16(R, W, G)                                                                                                   \
    do {                                                                                                                 \
        const double ms = time_ms([&] { hipLaunchKernelGGL((mix16<R, W>), dim3(G), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, n16); }, reps); \
        printf("16B/lane  read %d : write %d  grid %-6d %8.3f %9.2f\n", R, W, G, ms, (R + W) * (double)stream_bytes / ms / 1e9);     \
    } while (0)
#define RUN4(R, W, G)                                                                                                    \
    do {                                                                                                                 \
        const double ms = time_ms([&] { hipLaunchKernelGGL((mix4<R, W>), dim3(G), dim3(256), 0, 0, (const unsigned*)src, (unsigned*)dst, n4); }, reps); \
        printf(" 4B/lane  read %d : write %d  grid %-6d %8.3f %9.2f\n", R, W, G, ms, (R + W) * (double)stream_bytes / ms / 1e9);     \
    } while (0)
    for (int G : {2048, 8192, 32768}) {
        RUN16(1, 0, G); RUN16(2, 0, G); RUN16(0, 1, G); RUN16(0, 2, G); RUN16(1, 1, G); RUN16(2, 1, G); RUN16(1, 2, G); RUN16(1, 3, G); RUN16(3, 1, G);
        RUN16(2, 4, G);
    }
    for (int G : {8192, 32768}) {
        RUN4(0, 1, G); RUN4(0, 3, G); RUN4(1, 3, G); RUN4(1, 1, G); RUN4(2, 6, G);
    }
    return 0;
}
