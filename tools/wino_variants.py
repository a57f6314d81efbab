#!/usr/bin/env python
"""What bounds the Winograd transforms?  Builds ablated copies of csrc/winograd.hip ON THE GPU BOX (hipcc, ~15 s each; the other
objects of the in-tree build are re-linked unchanged) and times the plane-writing input transform and the output transform with
each: full kernel / stores suppressed / loads suppressed / non-temporal stores.  Ablated variants compute garbage by
construction -- this is a timing probe (tools/, not product).

    python tools/wino_variants.py [--reps 10]"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bbdm_amd", "csrc")
SRC = open(os.path.join(CSRC, "winograd.hip")).read()

IN_STORE = """            *reinterpret_cast<unsigned*>(o) = p1;
            *reinterpret_cast<unsigned*>(o + 1024) = p2;
            *reinterpret_cast<unsigned*>(o + 2048) = p3;
            o += plane;
        }
        __builtin_amdgcn_sched_barrier(0);"""
IN_LOAD = "            d[i] = *reinterpret_cast<const float2*>(x + ((size_t)(n * Hs + hs) * Ws + wsrc) * ldx + c);\n        }\n#pragma unroll\n        for (int i = 0; i < AL; ++i) {\n            const int h = MO * th - 1 + i;\n            const float mask = (h >= 0 && h < H) ? wmask : 0.f;\n            float2 v = d[i];\n            if (PRE) {\n                v.x = v.x * s2.x + b2.x; v.y = v.y * s2.y + b2.y;\n                if (pre_silu) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); }\n            }\n            d[i] = make_float2(mask * v.x, mask * v.y);\n        }\n        bt_transform<MO>(d, col);\n#pragma unroll\n        for (int i = 0; i < AL; ++i) t[i][jj] = col[i];"


def in_variant(kind):
    s = SRC
    assert IN_STORE in s and IN_LOAD in s, "csrc/winograd.hip changed: update the patch anchors"
    if kind == "no_store":          # keep the arithmetic alive: store only for a value that never occurs
        s = s.replace(IN_STORE, IN_STORE.replace("            *reinterpret_cast<unsigned*>(o) = p1;",
                                                 "            if (p1 == 0x7fc12345u && p2 == 0x7fc54321u) *reinterpret_cast<unsigned*>(o) = p1;")
                      .replace("            *reinterpret_cast<unsigned*>(o + 1024) = p2;\n", "")
                      .replace("            *reinterpret_cast<unsigned*>(o + 2048) = p3;\n", "            if (p3 == 0x7fc99999u) *reinterpret_cast<unsigned*>(o + 2048) = p3;\n"))
    elif kind == "no_load":
        s = s.replace(IN_LOAD, IN_LOAD.replace("d[i] = *reinterpret_cast<const float2*>(x + ((size_t)(n * Hs + hs) * Ws + wsrc) * ldx + c);",
                                               "d[i] = make_float2((float)(hs + wsrc) * 1e-3f, (float)(c + n) * 1e-3f);"))
    elif kind == "nt_store":
        s = s.replace(IN_STORE, IN_STORE.replace("*reinterpret_cast<unsigned*>(o) = p1;", "__builtin_nontemporal_store(p1, reinterpret_cast<unsigned*>(o));")
                      .replace("*reinterpret_cast<unsigned*>(o + 1024) = p2;", "__builtin_nontemporal_store(p2, reinterpret_cast<unsigned*>(o + 1024));")
                      .replace("*reinterpret_cast<unsigned*>(o + 2048) = p3;", "__builtin_nontemporal_store(p3, reinterpret_cast<unsigned*>(o + 2048));"))
    elif kind.startswith("lb"):     # min waves per SIMD -> VGPR cap 512 / n
        s = s.replace("__global__ void __launch_bounds__(256) winograd_input_split_kernel", f"__global__ void __launch_bounds__(256, {kind[2:]}) winograd_input_split_kernel")
    elif kind != "full":
        raise ValueError(kind)
    assert kind == "full" or s != SRC
    return s


OUT_LOAD = "            for (int i = 0; i < AL; ++i) v[i] = *reinterpret_cast<const float2*>(m + (size_t)(i * AL + j) * plane);"
OUT_STORE = "                    *reinterpret_cast<float2*>(y + ((size_t)(n * H + oh) * W + ow) * ldy + c) = val;"


def out_variant(kind):
    s = SRC
    assert OUT_LOAD in s and OUT_STORE in s, "csrc/winograd.hip changed: update the patch anchors"
    if kind == "no_store":
        s = s.replace(OUT_STORE, "                    if (val.x == 1.2345e33f) *reinterpret_cast<float2*>(y + ((size_t)(n * H + oh) * W + ow) * ldy + c) = val;")
    elif kind == "no_load":
        s = s.replace(OUT_LOAD, "            for (int i = 0; i < AL; ++i) v[i] = make_float2((float)(i + j + c) * 1e-3f, (float)(tw + th) * 1e-3f);")
    elif kind == "nt_load":
        s = s.replace(OUT_LOAD, "            for (int i = 0; i < AL; ++i) { const float* q = m + (size_t)(i * AL + j) * plane; "
                                "v[i] = make_float2(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1)); }")
    elif kind.startswith("lb"):
        s = s.replace("__global__ void __launch_bounds__(256) winograd_output6_kernel", f"__global__ void __launch_bounds__(256, {kind[2:]}) winograd_output6_kernel")
    elif kind != "full":
        raise ValueError(kind)
    assert kind == "full" or s != SRC
    return s


def build(name, text):
    d = f"/tmp/winov/{name}"
    os.makedirs(d, exist_ok=True)
    open(f"{d}/winograd.hip", "w").write(text)
    others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if os.path.basename(o) != "winograd.o"]
    so = f"{d}/lib.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", CSRC, "-c",
                           f"{d}/winograd.hip", "-o", f"{d}/winograd.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", f"{d}/winograd.o"] + others + ["-o", so])
    return ctypes.CDLL(so)


def _time(fn, reps):
    assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default=None, help="comma-separated variant names to run (default: all)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    P_, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    shapes = [(16, 64, 64, 1024), (16, 256, 256, 128), (16, 128, 128, 512)]
    m, P = 6, 64
    for side, kinds, variant in (("input (writes 3 bf16 planes)", ("full", "no_store", "no_load", "nt_store", "lb3"), in_variant),
                                 ("output (+ residual, + GroupNorm statistics)", ("full", "no_store", "no_load", "nt_load", "lb3", "lb4"), out_variant)):
        print(side)
        for kind in kinds:
            if args.only and kind not in args.only.split(","):
                continue
            lib = build(("in_" if side.startswith("input") else "out_") + kind, variant(kind))
            lib.bbdm_winograd_tiles.restype = ctypes.c_size_t
            lib.bbdm_winograd_input_bf3p_f32.argtypes = [I, P_, I, P_, P_, P_, I, I, I, I, I, I, I, P_]
            lib.bbdm_winograd_output_stats_f32.argtypes = [I, P_, P_, P_, I, P_, I, I, I, I, I, I, P_, I, I, P_, I, I, P_]
            line = f"  {kind:9s}"
            for N, H, W, C in shapes:
                tiles = lib.bbdm_winograd_tiles(m, N, H, W)
                x = torch.randn(N, H, W, C, device=dev)
                if side.startswith("input"):
                    sc, bi = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
                    Vp = torch.empty(P * tiles * C * 6, dtype=torch.uint8, device=dev)
                    fn = lambda: lib.bbdm_winograd_input_bf3p_f32(m, x.data_ptr(), C, Vp.data_ptr(), sc.data_ptr(), bi.data_ptr(), C, 1, 0,
                                                                  N, H, W, C, st)
                    gb = (x.numel() * 4 + P * tiles * C * 6) / 1e9
                else:
                    M = torch.randn(P * tiles * C, device=dev)
                    b = torch.randn(C, device=dev)
                    y = torch.empty_like(x)
                    stats = torch.zeros(N * 64, dtype=torch.float64, device=dev)
                    fn = lambda: lib.bbdm_winograd_output_stats_f32(m, M.data_ptr(), b.data_ptr(), x.data_ptr(), C, y.data_ptr(), C, 0, N, H,
                                                                    W, C, stats.data_ptr(), max(4, C // 32), 0, None, 0, 0, st)
                    gb = (P * tiles * C * 4 + 2 * x.numel() * 4) / 1e9
                ms = _time(fn, args.reps)
                line += f" | N{N} {H}x{W} C{C}: {ms:6.3f} ms ({gb / ms:5.2f} TB/s of the full kernel's bytes)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
