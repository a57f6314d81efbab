#!/usr/bin/env python
"""The Winograd input transform + tile GEMMs on the two bf16x3 paths, per layer shape of the C2 step:

  old: bbdm_winograd_input_f32 (fp32 V)            -> bbdm_winograd_gemm_bf3_f32  (csrc/gemm_bf3.hip: V split while staged)
  new: bbdm_winograd_input_bf3p_f32 (3 bf16 planes) -> bbdm_winograd_gemm_bf3p_f32 (csrc/gemm_bf3p.hip: LDS-DMA copies + MFMAs)
       for every kernel of gemm_bf3p.hip (option bf3p_kernel: forced tile shapes 5, 4 and the default 6)

    python tools/bf3p_bench.py [--reps 10] [--kernels 5,4,6]
Prints ms and fp32-equivalent TFLOP/s (HIP events on the launch stream), checks that M is bit-equal between the paths, and a
launch-weighted C2 total (weights = how often the shape occurs in the 256^2 / batch-16 step)."""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib
import kernel_ops as ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, launches per C2 step (profiles/r02_c2_per_launch.md: all 42 Winograd layers)
    (16, 64, 64, 1024, 1024, 10),
    (16, 256, 256, 128, 128, 7),
    (16, 128, 128, 512, 512, 6),
    (16, 64, 64, 512, 512, 2),
    (16, 64, 64, 2048, 1024, 2),
    (16, 256, 256, 512, 512, 2),
    (16, 256, 256, 256, 128, 2),
    (16, 128, 128, 128, 128, 2),
    (16, 128, 128, 1024, 1024, 2),
    (16, 64, 64, 512, 1024, 1),
    (16, 64, 64, 1536, 1024, 1),
    (16, 256, 256, 640, 128, 1),
    (16, 128, 128, 640, 512, 1),
    (16, 128, 128, 1536, 512, 1),
    (16, 128, 128, 128, 512, 1),
    (16, 128, 128, 1024, 512, 1),
]


C3_SHAPES = [  # m, N, H, W, Cin, Cout, launches per C3 step (LBBDM-f4 latent 64x64, batch 32: gpurun_out/r03q/c3_per_launch.md)
    (4, 32, 16, 16, 1024, 1024, 10),
    (4, 32, 32, 32, 1024, 1024, 2),
    (4, 32, 32, 32, 512, 512, 6),
    (6, 32, 64, 64, 512, 512, 2),
    (4, 32, 16, 16, 2048, 1024, 2),
    (4, 32, 32, 32, 1536, 512, 1),
    (6, 32, 64, 64, 128, 128, 7),
    (4, 32, 32, 32, 1024, 512, 1),
    (4, 32, 16, 16, 1536, 1024, 1),
    (4, 32, 32, 32, 640, 512, 1),
    (6, 32, 64, 64, 640, 128, 1),
    (6, 32, 64, 64, 256, 128, 2),
]


def _time(fn, reps):
    fn()
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
    ap.add_argument("--m", type=int, default=6)
    ap.add_argument("--kernels", default="5,4,6")
    ap.add_argument("--shapes", default=None, help="indices into SHAPES, comma separated")
    ap.add_argument("--set", default="c2", choices=("c2", "c3"), help="layer shapes of the C2 step (all m = --m) or of the C3 step")
    args = ap.parse_args()
    kernels = [int(k) for k in args.kernels.split(",")]
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    tot = {"in_old": 0.0, "in_new": 0.0, "g_old": 0.0, **{f"g{k}": 0.0 for k in kernels}}
    tot_fl = 0.0
    allshapes = [(args.m,) + s for s in SHAPES] if args.set == "c2" else C3_SHAPES
    shapes = allshapes if args.shapes is None else [allshapes[int(i)] for i in args.shapes.split(",")]
    for m, N, H, W, Cin, Cout, cnt in shapes:
        P = (m + 2) ** 2
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        x = torch.randn(N, H, W, Cin, device=dev)
        sc = torch.rand(N, Cin, device=dev) + 0.5
        bi = torch.randn(N, Cin, device=dev) * 0.1
        V = torch.empty(P * tiles * Cin, device=dev)
        Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        M0 = torch.empty(P * tiles * Cout, device=dev)
        M1 = torch.empty(P * tiles * Cout, device=dev)
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(P, Cin, Cout), dtype=torch.int16, device=dev)
        Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3_pack_f32", pw.data_ptr(), pk.data_ptr(), P, Cin, Cout, st)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
        in_old = lambda: _lib.call("bbdm_winograd_input_f32", m, x.data_ptr(), Cin, V.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin,
                                   1, 0, N, H, W, Cin, st)
        in_new = lambda: _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, Vp.data_ptr(), sc.data_ptr(), bi.data_ptr(),
                                   Cin, 1, 0, N, H, W, Cin, st)
        g_old = lambda: _lib.call("bbdm_winograd_gemm_bf3_f32", m, V.data_ptr(), pk.data_ptr(), M0.data_ptr(), N, H, W, Cin, Cout, st)
        g_new = lambda: _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M1.data_ptr(), N, H, W, Cin, Cout, st)
        t_in_old, t_in_new = _time(in_old, args.reps), _time(in_new, args.reps)
        t_old = _time(g_old, args.reps)
        fl = 2.0 * P * tiles * Cin * Cout
        T_raw = N * -(-H // m) * -(-W // m)
        line = (f"F{m} N{N} {H}x{W} {Cin}->{Cout} x{cnt}: input {t_in_old:6.3f} -> {t_in_new:6.3f} ms | gemm_bf3 {t_old:6.3f} ms "
                f"{fl / t_old / 1e9:6.1f} TF |")
        tot["in_old"] += cnt * t_in_old
        tot["in_new"] += cnt * t_in_new
        tot["g_old"] += cnt * t_old
        tot_fl += cnt * fl
        for k in kernels:
            lib.bbdm_set_option(b"bf3p_kernel", k)
            M1.zero_()
            t = _time(g_new, args.reps)
            eq = torch.equal(M0.view(P, tiles, Cout)[:, :T_raw], M1.view(P, tiles, Cout)[:, :T_raw])
            tot[f"g{k}"] += cnt * t
            line += f" k{k} {t:6.3f} ms {fl / t / 1e9:6.1f} TF {'==' if eq else '!= MISMATCH'} |"
        lib.bbdm_set_option(b"bf3p_kernel", 6)
        print(line, flush=True)
        del x, V, Vp, M0, M1, pw, pk, Bp
    print("C2-weighted totals (ms per step): " + "  ".join(f"{k} {v:.2f}" for k, v in tot.items()))
    print("C2-weighted GEMM TFLOP/s fp32-eq: " + "  ".join(
        f"{k} {tot_fl / v / 1e9:.1f} (frac {tot_fl / v / 1e9 / 416.67:.3f})" for k, v in tot.items() if k.startswith("g")))


if __name__ == "__main__":
    main()
