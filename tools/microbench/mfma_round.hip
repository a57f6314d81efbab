// mfma_round.hip -- what v_mfma_f32_32x32x16_{f16,bf16} does with its sum: rounding of C + sum(products), fp16 denormal inputs.
// Build: hipcc -O2 --offload-arch=gfx950 tools/microbench/mfma_round.hip -o tools/microbench/mfma_round ; run on the GPU box.
// Every lane supplies the same 8 values: A[i][k] = a[k % 8], B[k][j] = b[k % 8]; D = c + 2 * sum_e a[e] b[e].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_f16(const float* a, const float* b, float c, float* out) {
    f16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = (_Float16)a[e]; vb[e] = (_Float16)b[e]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void k_bf16(const float* a, const float* b, float c, float* out) {
    bf16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = (__bf16)a[e]; vb[e] = (__bf16)b[e]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
// fp32 -> fp16 pair split as the GEMM producers do it: what does the conversion do with denormal results?
__global__ void k_cvt(const float* x, float* out) {
    const float v = x[threadIdx.x];
    const _Float16 h = (_Float16)v;
    const float r = v - (float)h;
    const _Float16 h2 = (_Float16)r;
    out[2 * threadIdx.x] = (float)h;
    out[2 * threadIdx.x + 1] = (float)h2;
}

static float run(bool f16, const float* a, const float* b, float c) {
    float *da, *db, *dout, out;
    hipMalloc(&da, 32); hipMalloc(&db, 32); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 32, hipMemcpyHostToDevice); hipMemcpy(db, b, 32, hipMemcpyHostToDevice);
    if (f16) hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    else hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return out;
}
static void show(const char* what, float got, double exact) {
    const float rne = (float)exact;
    const float rtz = (fabs((double)rne) > fabs(exact)) ? nextafterf(rne, 0.f) : rne;
    printf("%-58s got %.9g (%a)  exact %.12g  RNE %a  RTZ %a  -> %s\n", what, got, got, exact, rne, rtz,
           got == rne && got != rtz ? "RNE" : got == rtz && got != rne ? "RTZ" : got == rne ? "rne==rtz" : "OTHER");
}
int main() {
    for (int f16 = 1; f16 >= 0; --f16) {
        printf("---- %s ----\n", f16 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_32x32x16_bf16");
        float a[8], b[8];
        // (1) c = 1, sum of 16 equal products = fr ulp(1) (ulp = 2^-23)
        const double frs[] = {0.375, 0.5, 0.625, 0.75, 1.25, 1.5, -0.375, -0.5, -0.625, -0.25, -0.75};
        for (double fr : frs) {
            // product p = fr * 2^-23 / 16: a = 2^-12, b = fr * 2^-15  (both exact in fp16 / bf16 for these fr)
            for (int e = 0; e < 8; ++e) { a[e] = ldexpf(1.f, -12); b[e] = (float)(fr * ldexp(1.0, -15)); }
            char s[96]; snprintf(s, sizeof s, "c=1, 16 products of %+.4g/16 ulp", fr);
            show(s, run(f16, a, b, 1.0f), 1.0 + fr * ldexp(1.0, -23));
        }
        // (2) c = 1 + 2^-23 (odd mantissa), tie cases
        for (double fr : {0.5, -0.5}) {
            for (int e = 0; e < 8; ++e) { a[e] = ldexpf(1.f, -12); b[e] = (float)(fr * ldexp(1.0, -15)); }
            char s[96]; snprintf(s, sizeof s, "c=1+ulp, sum = %+.4g ulp (tie)", fr);
            show(s, run(f16, a, b, 1.0f + ldexpf(1.f, -23)), 1.0 + ldexp(1.0, -23) + fr * ldexp(1.0, -23));
        }
        // (3) products of mixed sign and magnitude that cancel to a small residual: is the sum of products exact before C is added?
        {
            const float pa[8] = {1024.f, -1024.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float pb[8] = {1.f, 1.f, ldexpf(1.f, -13), 0.f, 0.f, 0.f, 0.f, 0.f};       // 1024 - 1024 + 2^-13 (x 2 halves)
            show("c=0, 1024 - 1024 + 2^-13 (x2)", run(f16, pa, pb, 0.f), 2 * ldexp(1.0, -13));
            show("c=2^-30, same", run(f16, pa, pb, ldexpf(1.f, -30)), 2 * ldexp(1.0, -13) + ldexp(1.0, -30));
            const float qa[8] = {1024.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            const float qb[8] = {1024.f, ldexpf(1.f, -6), ldexpf(1.f, -6), ldexpf(1.f, -6), ldexpf(1.f, -6), ldexpf(1.f, -6), ldexpf(1.f, -6), ldexpf(1.f, -6)};
            // 2^20 + 7 * 2^-6 per half: 2^21 + 14 * 2^-6 = 2^21 + 0.21875: ulp(2^21) = 0.25 -> 0.875 ulp
            show("c=0, 2*(2^20 + 7*2^-6): small products vs a large one", run(f16, qa, qb, 0.f), 2 * (1048576.0 + 7 * ldexp(1.0, -6)));
        }
        // (4) denormal inputs (fp16: below 2^-14; bf16: below 2^-126, skipped)
        if (f16) {
            for (int e = 0; e < 8; ++e) { a[e] = ldexpf(1.f, -20); b[e] = 1024.f; }
            show("c=0, a = 2^-20 (fp16 denormal), b = 2^10, x16", run(1, a, b, 0.f), 16 * ldexp(1.0, -10));
            for (int e = 0; e < 8; ++e) { a[e] = ldexpf(3.f, -24); b[e] = 1.f; }
            show("c=0, a = 3 * 2^-24 (fp16 denormal), b = 1, x16", run(1, a, b, 0.f), 16 * 3 * ldexp(1.0, -24));
        }
    }
    // conversions
    float hx[64], hout[128], *dx, *dout;
    for (int i = 0; i < 64; ++i) hx[i] = ldexpf(1.f + (float)i / 64.f + ldexpf(1.f, -20), -i / 2);
    hipMalloc(&dx, sizeof hx); hipMalloc(&dout, sizeof hout);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, dout);
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    printf("---- fp32 -> fp16 + fp16 residual ----\n");
    for (int i = 0; i < 64; i += 3) printf("x=%a  h1=%a  h2=%a  x-h1-h2=%a (rel %.2e)\n", hx[i], hout[2 * i], hout[2 * i + 1],
                                            hx[i] - hout[2 * i] - hout[2 * i + 1], (hx[i] - hout[2 * i] - hout[2 * i + 1]) / hx[i]);
    return 0;
}
