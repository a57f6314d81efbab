#!/usr/bin/env python
"""Where does gemm_bf3_kernel's time go?  Builds ablated copies of csrc/gemm_bf3.hip ON THE GPU BOX (hipcc, seconds each) and
times the same GEMM with each: full kernel / fragments never re-read from LDS / no staging (no global loads, split or LDS
stores) / no barrier.  Results are wrong by construction in the ablated variants -- this is a timing probe (tools/, not product)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bbdm_amd", "csrc")
src = open(os.path.join(CSRC, "gemm_bf3.hip")).read()

LOOP_OLD = src[src.index("    load(0);\n    store(smem);\n    __syncthreads();\n    for (int chunk = 0;"):src.index("    // ---- epilogue: + bias (+ residual)")]


def variant(no_lds_read=False, no_stage=False, no_barrier=False, no_mfma=False):
    loop = LOOP_OLD
    if no_stage:
        loop = loop.replace("        if (more) load(chunk + 1);\n", "").replace(
            "        if (more) store(smem + ((chunk + 1) & 1) * STAGE);\n", "")
    if no_barrier:
        loop = loop.replace("        if (more) store(smem + ((chunk + 1) & 1) * STAGE);\n        __syncthreads();\n",
                            "        if (more) store(smem + ((chunk + 1) & 1) * STAGE);\n") if not no_stage else \
            loop.replace("        __syncthreads();\n    }\n", "    }\n")
    if no_lds_read:
        loop = loop.replace("        const unsigned char* st = smem + (chunk & 1) * STAGE;", "        const unsigned char* st = smem;")
        loop = loop.replace("        bf16x8 af[2][3], bf[2][3];\n#pragma unroll\n        for (int t = 0; t < 2; ++t)\n#pragma unroll\n            for (int p = 0; p < 3; ++p) {",
                            "        if (chunk == 0)\n#pragma unroll\n        for (int t = 0; t < 2; ++t)\n#pragma unroll\n            for (int p = 0; p < 3; ++p) {")
        loop = loop.replace("    for (int chunk = 0; chunk < a.nchunks; ++chunk) {", "    bf16x8 af[2][3], bf[2][3];\n    for (int chunk = 0; chunk < a.nchunks; ++chunk) {")
    if no_mfma:
        loop = loop.replace("acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][BF3_TA[t]], bf[j][BF3_TB[t]], acc[i][j], 0, 0, 0);",
                            "acc[i][j][t] += (float)af[i][BF3_TA[t]][0] * (float)bf[j][BF3_TB[t]][0];")
    assert loop != LOOP_OLD or not (no_lds_read or no_stage or no_barrier or no_mfma), "a patch did not apply: csrc/gemm_bf3.hip changed"
    return src.replace(LOOP_OLD, loop)


def build(name, text):
    d = f"/tmp/bf3v/{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/gemm_bf3.hip", "w").write(text)
    so = f"{d}/lib.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-I", CSRC,
                           f"{d}/gemm_bf3.hip", os.path.join(CSRC, "runtime.hip"), "-o", so])
    return ctypes.CDLL(so)


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    variants = {"full": {}, "no_lds_read": dict(no_lds_read=True), "no_stage": dict(no_stage=True),
                "no_stage+no_lds_read": dict(no_stage=True, no_lds_read=True), "no_barrier": dict(no_barrier=True),
                "no_mfma": dict(no_mfma=True)}
    shapes = [(262144, 1024, 1024), (262144, 512, 512)]
    for name, kw in variants.items():
        lib = build(name, variant(**kw))
        P, L = ctypes.c_void_p, ctypes.c_longlong
        lib.bbdm_conv1x1_bf3_f32.argtypes = [P, ctypes.c_int, P, P, P, ctypes.c_int, P, ctypes.c_int, L, ctypes.c_int, ctypes.c_int, P]
        lib.bbdm_gemm_bf3_packed_halfs.restype = ctypes.c_size_t
        for pixels, Cin, Cout in shapes:
            x = torch.randn(pixels, Cin, device=dev)
            pk = torch.randint(-30000, 30000, (lib.bbdm_gemm_bf3_packed_halfs(1, Cin, Cout),), dtype=torch.int16, device=dev) & 0x3FFF
            out = torch.empty(pixels, Cout, device=dev)
            call = lambda: lib.bbdm_conv1x1_bf3_f32(x.data_ptr(), Cin, pk.data_ptr(), None, None, 0, out.data_ptr(), Cout, pixels,
                                                    Cin, Cout, st)
            assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"{name:24s} {pixels}x{Cin}->{Cout}: {ms:7.3f} ms  {2.0 * pixels * Cin * Cout / ms / 1e9:6.1f} TFLOP/s fp32-eq", flush=True)
            del x, pk, out


if __name__ == "__main__":
    main()
