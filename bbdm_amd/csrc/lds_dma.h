// lds_dma.h -- the LDS-DMA copy (global_load_lds_dwordx4) and its counted wait, shared by gemm_bf3p.hip and attention.hip.
#pragma once
#include "common.h"

// all but the newest N of this wave's copies (and global loads) have landed
#ifndef BBDM_WAIT_VMCNT           // (tools/hipemu/hip/hip_runtime.h defines the CPU emulator's form)
#define BBDM_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#endif
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    BBDM_WAIT_VMCNT(N);
}

// one LDS-DMA: every lane fetches 16 B at unit + lane16 (unit: wave-uniform, pinned to SGPRs so that the instruction takes its
// scalar-base + 32-bit-lane-offset form instead of a 64-bit address VGPR pair per copy), the wave's 1 KB lands at
// lds_wave_base + lane * 16
static __device__ __forceinline__ void glds16(const unsigned char* unit, unsigned lane16, unsigned char* lds_wave_base) {
    const unsigned long long u = (unsigned long long)(uintptr_t)unit;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    const unsigned char* base = (const unsigned char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + lane16),
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

