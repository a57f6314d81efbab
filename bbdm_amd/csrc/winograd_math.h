// winograd_math.h -- the 1-D Winograd transforms shared by winograd.hip (forward / data gradient) and winograd_wgrad.hip
// (weight gradient).  Host + device: bbdm_debug_winograd_transform_1d runs the same code on the CPU for the tests
// (tests/test_winograd_math_cpu.py checks every hand-factored formula against its transform matrix).
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 operator*(float s, float2 a) { return make_float2(s * a.x, s * a.y); }

// ---- the 1-D transforms (host + device: bbdm_debug_winograd_transform_1d runs the same code on the CPU for the tests) ---
// t = B^T d.  m = 2: points {0, 1, -1, inf}; m = 4: points {0, 1, -1, 2, -2, inf} (Lavin & Gray, arXiv:1509.09308).
template <int MO, typename T>
__host__ __device__ __forceinline__ void bt_transform(const T (&d)[MO + 2], T (&t)[MO + 2]) {
    if constexpr (MO == 2) {
        t[0] = d[0] - d[2];
        t[1] = d[1] + d[2];
        t[2] = d[2] - d[1];
        t[3] = d[1] - d[3];
    } else if constexpr (MO == 4) {
        t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        t[1] = (d[3] + d[4]) - 4.f * (d[1] + d[2]);
        t[2] = 4.f * (d[1] - d[2]) + (d[4] - d[3]);
        t[3] = 2.f * (d[3] - d[1]) + (d[4] - d[2]);
        t[4] = 2.f * (d[1] - d[3]) + (d[4] - d[2]);
        t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    } else if constexpr (MO == 8) {
        // m = 8 (round 5): points {0, +-1/2, +-3/4, +-4/3, +-2, inf} -- the set with the smallest fp32 error of 330 candidates (the usual
        // {0, +-1, +-2, +-1/2, +-4}: 15x more; tests/test_winograd_math_cpu.py); rows scaled so that every entry is a dyadic rational, i.e.
        // EXACT in fp32 (and G, evaluated in fp64, carries the reciprocals): the algebra holds exactly, only the data round.
        t[0] = 0x1.2p-3f * d[0] - 0x1.da8p-1f * d[2] + 0x1.ae1p+0f * d[4] - 0x1.da8p-1f * d[6] + 0x1.2p-3f * d[8];
        {
            const T ev = (-0x1.2p-1f * d[2]) + 0x1.75p+0f * d[4] - 0x1.c88p-1f * d[6] + 0x1.2p-3f * d[8];
            const T od = (-0x1.2p-2f * d[1]) + 0x1.75p-1f * d[3] - 0x1.c88p-2f * d[5] + 0x1.2p-4f * d[7];
            t[1] = ev + od;
            t[2] = ev - od;
        }
        {
            const T ev = (-0x1p-2f * d[2]) + 0x1.34p+0f * d[4] - 0x1.b2p-1f * d[6] + 0x1.2p-3f * d[8];
            const T od = (-0x1.8p-3f * d[1]) + 0x1.cep-1f * d[3] - 0x1.458p-1f * d[5] + 0x1.bp-4f * d[7];
            t[3] = ev + od;
            t[4] = ev - od;
        }
        {
            const T ev = (-0x1.bp-4f * d[2]) + 0x1.458p-1f * d[4] - 0x1.cep-1f * d[6] + 0x1.8p-3f * d[8];
            const T od = (-0x1.2p-3f * d[1]) + 0x1.b2p-1f * d[3] - 0x1.34p+0f * d[5] + 0x1p-2f * d[7];
            t[5] = ev + od;
            t[6] = ev - od;
        }
        {
            const T ev = (-0x1.2p-4f * d[2]) + 0x1.c88p-2f * d[4] - 0x1.75p-1f * d[6] + 0x1.2p-2f * d[8];
            const T od = (-0x1.2p-3f * d[1]) + 0x1.c88p-1f * d[3] - 0x1.75p+0f * d[5] + 0x1.2p-1f * d[7];
            t[7] = ev + od;
            t[8] = ev - od;
        }
        t[9] = 0x1.2p-3f * d[1] - 0x1.da8p-1f * d[3] + 0x1.ae1p+0f * d[5] - 0x1.da8p-1f * d[7] + 0x1.2p-3f * d[9];
    } else {        // m = 6: points {0, 1, -1, 2, -2, 1/2, -1/2, inf} (the 8x8 transform of NNPACK / wincnn)
        const T e0 = (d[2] + d[6]) - 4.25f * d[4], o0 = (d[1] + d[5]) - 4.25f * d[3];
        const T e1 = (d[6] + 0.25f * d[2]) - 1.25f * d[4], o1 = (0.5f * d[1] + 2.f * d[5]) - 2.5f * d[3];
        const T e2 = (d[6] + 4.f * d[2]) - 5.f * d[4], o2 = (2.f * d[1] + 0.5f * d[5]) - 2.5f * d[3];
        t[0] = (d[0] - d[6]) + 5.25f * (d[4] - d[2]);
        t[1] = e0 + o0;
        t[2] = e0 - o0;
        t[3] = e1 + o1;
        t[4] = e1 - o1;
        t[5] = e2 + o2;
        t[6] = e2 - o2;
        t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
    }
}
// s = A^T m
template <int MO, typename T>
__host__ __device__ __forceinline__ void at_transform(const T (&m)[MO + 2], T (&s)[MO]) {
    if constexpr (MO == 2) {
        s[0] = m[0] + m[1] + m[2];
        s[1] = m[1] - m[2] - m[3];
    } else if constexpr (MO == 4) {
        const T p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
        s[0] = m[0] + p12 + p34;
        s[1] = m12 + 2.f * m34;
        s[2] = p12 + 4.f * p34;
        s[3] = m12 + 8.f * m34 + m[5];
    } else if constexpr (MO == 8) {
        const T p0 = m[1] + m[2], q0 = m[1] - m[2], p1 = m[3] + m[4], q1 = m[3] - m[4], p2 = m[5] + m[6], q2 = m[5] - m[6],
                p3 = m[7] + m[8], q3 = m[7] - m[8];
        s[0] = m[0] + p0 + p1 + 0x1.116p-3f * p2 + 0x1p-7f * p3;
        s[1] = 0x1p-1f * q0 + 0x1.8p-1f * q1 + 0x1.6c8p-3f * q2 + 0x1p-6f * q3;
        s[2] = 0x1p-2f * p0 + 0x1.2p-1f * p1 + 0x1.e6p-3f * p2 + 0x1p-5f * p3;
        s[3] = 0x1p-3f * q0 + 0x1.bp-2f * q1 + 0x1.44p-2f * q2 + 0x1p-4f * q3;
        s[4] = 0x1p-4f * p0 + 0x1.44p-2f * p1 + 0x1.bp-2f * p2 + 0x1p-3f * p3;
        s[5] = 0x1p-5f * q0 + 0x1.e6p-3f * q1 + 0x1.2p-1f * q2 + 0x1p-2f * q3;
        s[6] = 0x1p-6f * p0 + 0x1.6c8p-3f * p1 + 0x1.8p-1f * p2 + 0x1p-1f * p3;
        s[7] = 0x1p-7f * q0 + 0x1.116p-3f * q1 + q2 + q3 + m[9];
    } else {
        const T p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4], p56 = m[5] + m[6],
                m56 = m[5] - m[6];
        s[0] = m[0] + p12 + p34 + p56;
        s[1] = m12 + 2.f * m34 + 0.5f * m56;
        s[2] = p12 + 4.f * p34 + 0.25f * p56;
        s[3] = m12 + 8.f * m34 + 0.125f * m56;
        s[4] = p12 + 16.f * p34 + 0.0625f * p56;
        s[5] = m12 + 32.f * m34 + 0.03125f * m56 + m[7];
    }
}
// ---- F(7x7, 2x2) on the SAME eight points (round 5): a 2-tap filter needs m + 1 points, so the 8-point transform yields 7 outputs
// per tile instead of 6.  B^T depends on the points only: bt_transform<6> serves both; G (8 x 2) and A^T (7 x 8) are below.  Used for
// the four phase filters of conv3x3(nearest x2 (x)) (csrc/winograd.hip: each phase reads 2 x 2 pixels of x): 64 / 49 = 1.31
// multiplies per output instead of 64 / 36 = 1.78.  y[o] = g[0] d[o] + g[1] d[o + 1], o = 0 .. 6.
template <typename T>
__host__ __device__ __forceinline__ void at_transform7(const T (&m)[8], T (&s)[7]) {
    const T p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4], p56 = m[5] + m[6], m56 = m[5] - m[6];
    s[0] = m[0] + p12 + p34 + p56;
    s[1] = m12 + 2.f * m34 + 0.5f * m56;
    s[2] = p12 + 4.f * p34 + 0.25f * p56;
    s[3] = m12 + 8.f * m34 + 0.125f * m56;
    s[4] = p12 + 16.f * p34 + 0.0625f * p56;
    s[5] = m12 + 32.f * m34 + 0.03125f * m56;
    s[6] = p12 + 64.f * p34 + 0.015625f * p56 + m[7];
}
__host__ __device__ __forceinline__ void g_transform72(const float (&g)[2], float (&u)[8]) {
    u[0] = g[0];
    u[1] = (-2.f / 9.f) * (g[0] + g[1]);
    u[2] = (-2.f / 9.f) * (g[0] - g[1]);
    u[3] = (1.f / 90.f) * g[0] + (1.f / 45.f) * g[1];
    u[4] = (1.f / 90.f) * g[0] - (1.f / 45.f) * g[1];
    u[5] = (32.f / 45.f) * g[0] + (16.f / 45.f) * g[1];
    u[6] = (32.f / 45.f) * g[0] - (16.f / 45.f) * g[1];
    u[7] = g[1];
}

// the type G / G^T are evaluated in: fp64 for m = 8, rounded to fp32 once
template <int MO> struct WinoWeightT { typedef float type; };
template <> struct WinoWeightT<8> { typedef double type; };

// u = G g.  (m = 8 is instantiated with T = double: its G holds ninths and 4125ths, and the transformed weights are rounded to fp32 once)
template <int MO, typename T = float>
__host__ __device__ __forceinline__ void g_transform(const T (&g)[3], T (&u)[MO + 2]) {
    if constexpr (MO == 2) {
        u[0] = g[0];
        u[1] = 0.5f * (g[0] + g[1] + g[2]);
        u[2] = 0.5f * (g[0] - g[1] + g[2]);
        u[3] = g[2];
    } else if constexpr (MO == 4) {
        u[0] = 0.25f * g[0];
        u[1] = (-1.f / 6.f) * (g[0] + g[1] + g[2]);
        u[2] = (-1.f / 6.f) * (g[0] - g[1] + g[2]);
        u[3] = (1.f / 24.f) * g[0] + (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
        u[4] = (1.f / 24.f) * g[0] - (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
        u[5] = g[2];
    } else if constexpr (MO == 8) {
        u[0] = (T)(64.0 / 9.0) * g[0];
        u[1] = (T)(-32768.0 / 4125.0) * g[0] + (T)(-16384.0 / 4125.0) * g[1] + (T)(-8192.0 / 4125.0) * g[2];
        u[2] = (T)(-32768.0 / 4125.0) * g[0] + (T)(16384.0 / 4125.0) * g[1] + (T)(-8192.0 / 4125.0) * g[2];
        u[3] = (T)(2097152.0 / 433125.0) * g[0] + (T)(524288.0 / 144375.0) * g[1] + (T)(131072.0 / 48125.0) * g[2];
        u[4] = (T)(2097152.0 / 433125.0) * g[0] + (T)(-524288.0 / 144375.0) * g[1] + (T)(131072.0 / 48125.0) * g[2];
        u[5] = (T)(-131072.0 / 48125.0) * g[0] + (T)(-524288.0 / 144375.0) * g[1] + (T)(-2097152.0 / 433125.0) * g[2];
        u[6] = (T)(-131072.0 / 48125.0) * g[0] + (T)(524288.0 / 144375.0) * g[1] + (T)(-2097152.0 / 433125.0) * g[2];
        u[7] = (T)(8192.0 / 4125.0) * g[0] + (T)(16384.0 / 4125.0) * g[1] + (T)(32768.0 / 4125.0) * g[2];
        u[8] = (T)(8192.0 / 4125.0) * g[0] + (T)(-16384.0 / 4125.0) * g[1] + (T)(32768.0 / 4125.0) * g[2];
        u[9] = (T)(64.0 / 9.0) * g[2];
    } else {
        u[0] = g[0];
        u[1] = (-2.f / 9.f) * (g[0] + g[1] + g[2]);
        u[2] = (-2.f / 9.f) * (g[0] - g[1] + g[2]);
        u[3] = (1.f / 90.f) * g[0] + (1.f / 45.f) * g[1] + (2.f / 45.f) * g[2];
        u[4] = (1.f / 90.f) * g[0] - (1.f / 45.f) * g[1] + (2.f / 45.f) * g[2];
        u[5] = (32.f / 45.f) * g[0] + (16.f / 45.f) * g[1] + (8.f / 45.f) * g[2];
        u[6] = (32.f / 45.f) * g[0] - (16.f / 45.f) * g[1] + (8.f / 45.f) * g[2];
        u[7] = g[2];
    }
}

// t = A s (the transpose of at_transform: m -> m+2; the dY side of the weight gradient, dM = A dY A^T)
template <int MO, typename T>
__host__ __device__ __forceinline__ void a_transform(const T (&s)[MO], T (&t)[MO + 2]) {
    if constexpr (MO == 2) {
        t[0] = s[0];
        t[1] = s[0] + s[1];
        t[2] = s[0] - s[1];
        t[3] = -1.f * s[1];
    } else if constexpr (MO == 4) {
        const T e = s[0] + s[2], o = s[1] + s[3], e2 = s[0] + 4.f * s[2], o2 = 2.f * s[1] + 8.f * s[3];
        t[0] = s[0];
        t[1] = e + o;
        t[2] = e - o;
        t[3] = e2 + o2;
        t[4] = e2 - o2;
        t[5] = s[3];
    } else if constexpr (MO == 8) {
        // the transpose of at_transform<8> (same dyadic coefficients, exact in fp32): point pairs +-1/2, +-3/4, +-4/3 (x (3/4)^7), +-2 (x 2^-7)
        const T e0 = ((s[0] + 0x1p-2f * s[2]) + 0x1p-4f * s[4]) + 0x1p-6f * s[6];
        const T o0 = ((0x1p-1f * s[1] + 0x1p-3f * s[3]) + 0x1p-5f * s[5]) + 0x1p-7f * s[7];
        const T e1 = ((s[0] + 0x1.2p-1f * s[2]) + 0x1.44p-2f * s[4]) + 0x1.6c8p-3f * s[6];
        const T o1 = ((0x1.8p-1f * s[1] + 0x1.bp-2f * s[3]) + 0x1.e6p-3f * s[5]) + 0x1.116p-3f * s[7];
        const T e2 = ((0x1.116p-3f * s[0] + 0x1.e6p-3f * s[2]) + 0x1.bp-2f * s[4]) + 0x1.8p-1f * s[6];
        const T o2 = ((0x1.6c8p-3f * s[1] + 0x1.44p-2f * s[3]) + 0x1.2p-1f * s[5]) + s[7];
        const T e3 = ((0x1p-7f * s[0] + 0x1p-5f * s[2]) + 0x1p-3f * s[4]) + 0x1p-1f * s[6];
        const T o3 = ((0x1p-6f * s[1] + 0x1p-4f * s[3]) + 0x1p-2f * s[5]) + s[7];
        t[0] = s[0];
        t[1] = e0 + o0;
        t[2] = e0 - o0;
        t[3] = e1 + o1;
        t[4] = e1 - o1;
        t[5] = e2 + o2;
        t[6] = e2 - o2;
        t[7] = e3 + o3;
        t[8] = e3 - o3;
        t[9] = s[7];
    } else {
        const T e1 = (s[0] + s[2]) + s[4], o1 = (s[1] + s[3]) + s[5];
        const T e2 = (s[0] + 4.f * s[2]) + 16.f * s[4], o2 = (2.f * s[1] + 8.f * s[3]) + 32.f * s[5];
        const T e3 = (s[0] + 0.25f * s[2]) + 0.0625f * s[4], o3 = (0.5f * s[1] + 0.125f * s[3]) + 0.03125f * s[5];
        t[0] = s[0];
        t[1] = e1 + o1;
        t[2] = e1 - o1;
        t[3] = e2 + o2;
        t[4] = e2 - o2;
        t[5] = e3 + o3;
        t[6] = e3 - o3;
        t[7] = s[5];
    }
}
// g = G^T u (the transpose of g_transform: m+2 -> 3; the last step of the weight gradient, dg = G^T dU G)
// (m = 8 is instantiated with T = double, like g_transform<8>: G holds ninths and 4125ths)
template <int MO, typename T = float>
__host__ __device__ __forceinline__ void gt_transform(const T (&u)[MO + 2], T (&g)[3]) {
    if constexpr (MO == 8) {
        const T p12 = u[1] + u[2], m12 = u[2] - u[1], p34 = u[3] + u[4], m34 = u[3] - u[4], p56 = u[5] + u[6], m56 = u[6] - u[5],
                p78 = u[7] + u[8], m78 = u[7] - u[8];
        g[0] = (T)(64.0 / 9.0) * u[0] - (T)(32768.0 / 4125.0) * p12 + (T)(2097152.0 / 433125.0) * p34 - (T)(131072.0 / 48125.0) * p56 +
               (T)(8192.0 / 4125.0) * p78;
        g[1] = (T)(16384.0 / 4125.0) * (m12 + m78) + (T)(524288.0 / 144375.0) * (m34 + m56);
        g[2] = (T)(64.0 / 9.0) * u[9] - (T)(8192.0 / 4125.0) * p12 + (T)(131072.0 / 48125.0) * p34 - (T)(2097152.0 / 433125.0) * p56 +
               (T)(32768.0 / 4125.0) * p78;
    } else if constexpr (MO == 2) {
        const float p = 0.5f * (u[1] + u[2]);
        g[0] = u[0] + p;
        g[1] = 0.5f * (u[1] - u[2]);
        g[2] = p + u[3];
    } else if constexpr (MO == 4) {
        const float p12 = u[1] + u[2], m12 = u[1] - u[2], p34 = u[3] + u[4], m34 = u[3] - u[4];
        g[0] = 0.25f * u[0] - (1.f / 6.f) * p12 + (1.f / 24.f) * p34;
        g[1] = (1.f / 12.f) * m34 - (1.f / 6.f) * m12;
        g[2] = (1.f / 6.f) * (p34 - p12) + u[5];
    } else {
        const float p12 = u[1] + u[2], m12 = u[1] - u[2], p34 = u[3] + u[4], m34 = u[3] - u[4], p56 = u[5] + u[6],
                    m56 = u[5] - u[6];
        g[0] = u[0] - (2.f / 9.f) * p12 + (1.f / 90.f) * p34 + (32.f / 45.f) * p56;
        g[1] = (1.f / 45.f) * m34 + (16.f / 45.f) * m56 - (2.f / 9.f) * m12;
        g[2] = (2.f / 45.f) * p34 + (8.f / 45.f) * p56 - (2.f / 9.f) * p12 + u[7];
    }
}

// max_i sum_j |B^T_ij| squared: |B^T d B| <= gain * max |d| for every transform point (the bound the fp16-pair planes are scaled by,
// h2_split.h; tests/test_winograd_math_cpu.py checks the constants against the matrices).  m = 7 shares m = 6's B^T.
__host__ __device__ inline float wino_input_gain(int m) {
    return m == 2 ? 4.f : m == 4 ? 100.f : m == 8 ? 21.f : 225.f;      // (m = 8: 20.955)
}

// tiles along one axis.  m = 7 (the phase-filter form, F(7x7, 2x2)): tile t covers the window rows 7 t - 1 .. 7 t + 6 of x; phase 0
// takes its 7 outputs at rows 7 t .. 7 t + 6, phase 1 at rows 7 t - 1 .. 7 t + 5 -- row H - 1 of phase 1 needs 7 t + 5 >= H - 1.
__host__ __device__ inline int wino_tdim(int H, int m) { return m == 7 ? (H + 7) / 7 : (H + m - 1) / m; }
inline size_t wino_tiles_raw(int N, int H, int W, int m) { return (size_t)N * wino_tdim(H, m) * wino_tdim(W, m); }
inline size_t wino_tiles_padded(int N, int H, int W, int m) {
    return (wino_tiles_raw(N, H, W, m) + 255) / 256 * 256;      // whole 8x32 GEMM tiles
}
inline int wino_planes(int m) { return m == 7 ? 64 : (m + 2) * (m + 2); }       // (m = 8: 100)

}  // namespace

#define BBDM_WINO_M(m) \
    BBDM_REQUIRE((m) == 2 || (m) == 4 || (m) == 6, "winograd: output tile m=%d unsupported (2, 4 or 6)", (m))
// ... entry points of the phase-filter form also take m = 7 = F(7x7, 2x2) on the 8-point transform (see wino_tdim)
#define BBDM_WINO_M7(m) \
    BBDM_REQUIRE((m) == 2 || (m) == 4 || (m) == 6 || (m) == 7, "winograd: output tile m=%d unsupported (2, 4, 6 or 7)", (m))
// ... and the forward entry points (pack / input / tile GEMMs / output; not the gradients) m = 8 = F(8x8, 3x3) on ten points
#define BBDM_WINO_M8(m) \
    BBDM_REQUIRE((m) == 2 || (m) == 4 || (m) == 6 || (m) == 8, "winograd: output tile m=%d unsupported (2, 4, 6 or 8)", (m))
#define BBDM_WINO_M78(m) \
    BBDM_REQUIRE((m) == 2 || (m) == 4 || (m) == 6 || (m) == 7 || (m) == 8, "winograd: output tile m=%d unsupported (2, 4, 6, 7 or 8)", (m))
#define BBDM_WINO_HW(m, H, W)                                                                                          \
    BBDM_REQUIRE((H) > 0 && (W) > 0 && ((m) >= 6 || ((H) % (m) == 0 && (W) % (m) == 0)),                                \
                 "winograd: H=%d, W=%d must be multiples of m=%d", H, W, m)
