#!/usr/bin/env python
"""Winograd layers of the C2 step two ways: input transform writing the three bf16 planes (6 B per element) + gemm_bf3p_pipe_kernel,
against input transform writing fp32 row units (4 B) + gemm_bf3q_pipe_kernel (every wave splits its share of A between its MFMAs).

    python tools/bf3q_bench.py [--reps 6]"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib
import kernel_ops as ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, launches per C2 step
    (16, 64, 64, 1024, 1024, 10), (16, 256, 256, 128, 128, 7), (16, 128, 128, 512, 512, 6), (16, 64, 64, 2048, 1024, 2),
    (16, 256, 256, 512, 512, 2), (16, 128, 128, 1024, 1024, 2), (16, 256, 256, 640, 128, 1), (16, 128, 128, 1536, 512, 1),
    (16, 256, 256, 256, 128, 2), (16, 128, 128, 128, 512, 1), (16, 64, 64, 512, 512, 2), (16, 64, 64, 1536, 1024, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    m, P = 6, 64
    tot = [0.0] * 4

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    for N, H, W, Cin, Cout, cnt in SHAPES:
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        x = torch.randn(N, H, W, Cin, device=dev)
        sc = torch.rand(N, Cin, device=dev) + 0.5
        bi = torch.randn(N, Cin, device=dev) * 0.1
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
        Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        Vf = torch.empty(lib.bbdm_gemm_bf3q_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        Mp = torch.empty(P * tiles * Cout, device=dev)
        Mq = torch.empty(P * tiles * Cout, device=dev)
        fns = [
            lambda: _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, Vp.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W, Cin, st),
            lambda: _lib.call("bbdm_winograd_input_bf3q_f32", m, x.data_ptr(), Cin, Vf.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W, Cin, st),
            lambda: _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), Mp.data_ptr(), N, H, W, Cin, Cout, st),
            lambda: _lib.call("bbdm_winograd_gemm_bf3q_f32", m, Vf.data_ptr(), Bp.data_ptr(), Mq.data_ptr(), N, H, W, Cin, Cout, st),
        ]
        t = [timed(f) for f in fns]
        same = torch.equal(Mp, Mq)
        flops = 2.0 * P * tiles * Cin * Cout
        for i in range(4):
            tot[i] += cnt * t[i]
        print(f"N{N} {H}x{W} {Cin}->{Cout} x{cnt}: input planes {t[0]:6.3f} units {t[1]:6.3f} ms | gemm bf3p {t[2]:6.3f} ({flops / t[2] / 1e9:5.1f} TF) "
              f"bf3q {t[3]:6.3f} ({flops / t[3] / 1e9:5.1f} TF) | {'bit-equal' if same else 'MISMATCH'}", flush=True)
        del x, pw, Bp, Vp, Vf, Mp, Mq
    print(f"C2-weighted: input planes {tot[0]:.2f} units {tot[1]:.2f} | gemm bf3p {tot[2]:.2f} bf3q {tot[3]:.2f} | "
          f"planes path {tot[0] + tot[2]:.2f} units path {tot[1] + tot[3]:.2f} ms")


if __name__ == "__main__":
    main()
