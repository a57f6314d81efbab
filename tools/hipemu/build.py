#!/usr/bin/env python
"""Build the HOST (CPU-emulated) twin of libbbdm_hip.so: tools/hipemu/_build/libbbdm_emu.so.  TEST INFRASTRUCTURE ONLY.

Every bbdm_amd/csrc/*.hip is lightly rewritten (dynamic-LDS declarations -> a pointer from the emulator, the few
`asm volatile` waits dropped / s_barrier -> __syncthreads) and compiled as plain C++ against tools/hipemu/hip/hip_runtime.h
with the ROCm clang (ext_vector_type).  The C-ABI and every line of kernel arithmetic / indexing are the shipped source.
"""
import glob
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "bbdm_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
LIB = os.path.join(OUT, "libbbdm_emu.so")

_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+|__align__\(\d+\)\s+)?((?:\w+\s+)*?\w+)\s+(\w+)\[\];")


def rewrite(src: str) -> str:
    src = _DYN.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(hipemu::dyn_smem());", src)
    src = re.sub(r'asm volatile\("s_barrier"[^;]*;', "__syncthreads();", src)
    # hand-written fragment reads: `asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_address), "n"(offset));`
    # with lds_address(p) = byte offset of p inside the emulated dynamic LDS allocation
    src = re.sub(r'asm volatile\("ds_read_b128 %0, %1 offset:%2"\s*:\s*"=v"\((.+?)\)\s*:\s*"v"\((.+?)\),\s*"n"\((.+?)\)\);',
                 r"\1 = *reinterpret_cast<const __typeof__(\1)*>((const char*)hipemu::dyn_smem() + (\2) + (\3));", src)
    src = re.sub(r'asm volatile\("ds_read_b128 %0, %1"\s*:\s*"=v"\((.+?)\)\s*:\s*"v"\((.+?)\)\);',
                 r"\1 = *reinterpret_cast<const __typeof__(\1)*>((const char*)hipemu::dyn_smem() + (\2));", src)
    src = re.sub(r'return \(unsigned\)\(uintptr_t\)\(__attribute__\(\(address_space\(3\)\)\) void\*\)p;',
                 "return (unsigned)((const char*)p - (const char*)hipemu::dyn_smem());", src)
    # counted waits for LDS-DMA copies: template form `"s_waitcnt vmcnt(%0)" ::"n"(N)` and literal form `"s_waitcnt vmcnt(3)"`
    src = re.sub(r'asm volatile\("s_waitcnt vmcnt\(%0\)"\s*::\s*"n"\((\w+)\)[^;]*;', r"hipemu::dma_wait(\1);", src)
    src = re.sub(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)"[^;]*;', r"hipemu::dma_wait(\1);", src)
    src = re.sub(r'asm volatile\("s_waitcnt[^;]*;', ";", src)
    return src


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime of the ROCm clang (LD_PRELOAD it into the python that loads the --asan build)."""
    return subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()


def build(force=False, asan=False) -> str:
    """``asan``: the same sources with -fsanitize=address,undefined into _build_asan/libbbdm_emu_asan.so (tools/run_asan_emu.sh;
    tests/kernel_ops.use_emulator() picks it when HIPEMU_ASAN=1).  CPU build only: no GPU sanitizer runs on this pool."""
    global OUT, LIB
    if asan:
        OUT, LIB = os.path.join(HERE, "_build_asan"), os.path.join(HERE, "_build_asan", "libbbdm_emu_asan.so")
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "*.cpp")) + \
        glob.glob(os.path.join(HERE, "hip", "*.h")) + [os.path.join(ROOT, "include", "bbdm_hip.h"), __file__]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps):
        return LIB
    flags = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unknown-pragmas",
             "-Wno-unknown-attributes", "-Wno-pass-failed", "-I", HERE, "-I", CSRC]
    san = ["-fsanitize=address,undefined", "-fno-sanitize=float-cast-overflow", "-fno-sanitize-recover=undefined", "-shared-libasan",
           "-fno-omit-frame-pointer", "-g", "-O1"] if asan else []
    flags += san
    objs, procs = [], []
    for s in srcs:
        base = os.path.splitext(os.path.basename(s))[0]
        cpp = os.path.join(OUT, base + ".emu.cpp")
        text = rewrite(open(s).read()).replace('#include "common.h"', f'#include "{os.path.join(CSRC, "common.h")}"')
        if not os.path.exists(cpp) or open(cpp).read() != text:
            open(cpp, "w").write(text)
        obj = os.path.join(OUT, base + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps if not d.endswith(".hip") or d == s):
            procs.append((s, subprocess.Popen([CLANG, "-x", "c++"] + flags + ["-c", cpp, "-o", obj])))
    rt = os.path.join(OUT, "hipemu_rt.o")
    procs.append(("hipemu.cpp", subprocess.Popen([CLANG] + flags + ["-c", os.path.join(HERE, "hipemu.cpp"), "-o", rt])))
    bad = [n for n, p in procs if p.wait() != 0]
    if bad:
        raise RuntimeError("hipemu build failed for " + ", ".join(bad))
    subprocess.check_call([CLANG, "-shared", "-fPIC"] + (["-fsanitize=address,undefined", "-shared-libasan"] if asan else []) +
                          objs + [rt, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, asan="--asan" in sys.argv))
