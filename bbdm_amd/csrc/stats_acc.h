// stats_acc.h -- exact, ORDER-INDEPENDENT accumulation of the GroupNorm statistics.
//
// The sum / sum of squares of one (image, group) is added up by hundreds of workgroups of the kernel that produces the tensor
// (csrc/winograd.hip, conv_igemm.hip; util.py:214-216 is what the sums feed).  Floating-point atomics would make the result depend
// on the order in which those workgroups -- and the waves inside them -- happen to finish: run-to-run differences in the last bits of
// an fp64 sum, which now and then flip the fp32 rounding of a normalisation coefficient and with it bits of the sample.  The reference
// pins its kernels with cudnn.deterministic (main.py:57-65); here the sums are made order-independent by construction:
//
//   an fp64 partial sum v is cut into three signed integer limbs of 40 value bits, v * 2^70 = l2 * 2^80 + l1 * 2^40 + l0 (+ the bits
//   of v below 2^-70, dropped -- a fixed function of v), and the limbs are added with 64-bit INTEGER atomics (LDS: ds_add_u64, HBM:
//   global_atomic_add_x2).  Integer addition is associative and commutative, so every order of the atomics leaves the same limbs;
//   a reader folds them back into one double in a fixed order.  Window: |v| < 2^50 with a resolution of 2^-70 (a per-thread partial
//   sum of squares of 1e15, or an activation of 1e-10, are both far outside what a UNet produces): every limb then holds at most 40
//   value bits, and the 23 spare bits take 8 M additions before a limb could wrap -- the top limb included (round 4 accepted |v| < 2^72,
//   which left the top limb one or two spare bits: a few huge partials would have wrapped it silently).  A non-finite or
//   out-of-window partial bumps a fourth word, and the reader returns NaN for that cell, as a diverged fp64 sum would have shown.
//   The same cells carry the training backward's order-dependent sums (GroupNorm / LayerNorm parameter gradients, bias gradients,
//   the loss): round 5, so that a training micro-step is bitwise reproducible too (the reference: cudnn.deterministic, main.py:57-65).
//
// Layout of one statistics slot: [N][G][2 (sum, sum of squares)][SA_W] 64-bit words, zeroed by the caller before the producers run.
#pragma once
#include <stdint.h>

constexpr int SA_W = 4;                    // words per accumulated value: limbs 0..2, then the count of unrepresentable partials

// v -> limbs (see above).  Every step is exact: t = v * 2^70 is a power-of-two scaling; h = trunc(t * 2^-80) takes t's leading bits, so
// t - h * 2^80 is the rest of t's significand, representable; likewise for the middle limb.
__device__ __forceinline__ bool sa_split(double v, long long (&l)[3]) {
    if (!(fabs(v) < 0x1p50)) {            // NaN, Inf, or outside the window
        l[0] = l[1] = l[2] = 0;
        return false;
    }
    const double t = v * 0x1p70;
    const double h = trunc(t * 0x1p-80);
    const double r1 = fma(-h, 0x1p80, t);
    const double m = trunc(r1 * 0x1p-40);
    const double r0 = fma(-m, 0x1p40, r1);
    l[2] = (long long)h;
    l[1] = (long long)m;
    l[0] = (long long)r0;                 // (truncates the bits of v below 2^-70)
    return true;
}

// cell += v, for a cell in LDS or in HBM (the address space is known after inlining: ds_add_u64 / global_atomic_add_x2)
__device__ __forceinline__ void sa_add(unsigned long long* cell, double v) {
    if (v == 0.0) return;
    long long l[3];
    if (sa_split(v, l)) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (l[i]) atomicAdd(cell + i, (unsigned long long)l[i]);
    } else {
        atomicAdd(cell + 3, 1ull);
    }
}

// cell += another cell's limbs (an LDS table flushed to HBM: no decomposition, the limbs add word by word)
__device__ __forceinline__ void sa_add_cell(unsigned long long* dst, const unsigned long long* src) {
#pragma unroll
    for (int i = 0; i < SA_W; ++i) {
        const unsigned long long w = src[i];
        if (w) atomicAdd(dst + i, w);
    }
}

// the accumulated value: limbs folded most-significant first (one fixed expression: the same limbs give the same double)
__device__ __forceinline__ double sa_fold(unsigned long long w0, unsigned long long w1, unsigned long long w2, unsigned long long bad) {
    if (bad) return __builtin_nan("");
    const double l0 = (double)(long long)w0, l1 = (double)(long long)w1, l2 = (double)(long long)w2;
    return fma(l2, 0x1p80, fma(l1, 0x1p40, l0)) * 0x1p-70;
}
__device__ __forceinline__ double sa_load(const unsigned long long* cell) { return sa_fold(cell[0], cell[1], cell[2], cell[3]); }

// Sum (s, q) over the lanes of a wave that share `seg`: the lanes of one segment are CONSECUTIVE (lane = channel or channel pair, seg =
// its GroupNorm group) and a segment is at most `maxlen` lanes long.  Segmented inclusive scan (Hillis-Steele over __shfl_up, a fixed
// order: the same inputs give the same bits); afterwards the LAST lane of every segment (`sa_seg_tail`) holds the segment's sums.
// Replaces same-address LDS atomics from all lanes of a group (serialised by the LDS, and order-dependent across waves).
// `width` (64 or 32): the scan runs inside aligned runs of that many lanes (32: the two halves of a wave hold the same channels).
__device__ __forceinline__ void sa_seg_scan2(double& s, double& q, int seg, int lane, int maxlen, int width = 64) {
    const int l = lane & (width - 1);
    int sd = __shfl_up(seg, 1);                                  // segment of lane - d
    for (int d = 1; d < maxlen; d <<= 1) {
        const double os = __shfl_up(s, d), oq = __shfl_up(q, d);
        if (l >= d && sd == seg) { s += os; q += oq; }
        sd = __shfl_up(seg, 2 * d);
    }
}
__device__ __forceinline__ bool sa_seg_tail(int seg, int lane, int width = 64) {
    const int nxt = __shfl_down(seg, 1);
    return (lane & (width - 1)) == width - 1 || nxt != seg;
}
