#!/usr/bin/env python
"""Direct implicit-GEMM 3x3 vs Winograd F(2x2,3x3) on the wide layers of the 256^2 / batch-16 step.

    python tools/wino_bench.py [--reps 5]
Prints ms per shape for both paths (HIP events on the launch stream) and the max relative difference."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbdm_amd import _lib, ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout
    (16, 64, 64, 1024, 1024),
    (16, 64, 64, 2048, 1024),
    (16, 64, 64, 512, 1024),
    (16, 128, 128, 512, 512),
    (16, 128, 128, 1024, 1024),
    (16, 128, 128, 1536, 512),
    (16, 128, 128, 128, 512),
    (16, 256, 256, 512, 512),
    (16, 256, 256, 640, 128),
    (16, 256, 256, 256, 128),
    (16, 256, 256, 128, 128),
    (8, 32, 32, 512, 512),
    (8, 16, 16, 512, 512),
    (8, 64, 64, 256, 256),
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
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    td = tw = 0.0
    for N, H, W, Cin, Cout in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
        b = torch.randn(Cout, device=dev)
        pd, pw = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
        o1 = torch.empty(N, H, W, Cout, device=dev)
        o2 = torch.empty(N, H, W, Cout, device=dev)
        ws = torch.empty(lib.bbdm_winograd_workspace_floats(N, H, W, Cin, Cout), device=dev)
        ms_d = _time(lambda: ops.conv2d_nhwc(x, pd, b, Cout, 3, out=o1), args.reps)
        ms_w = _time(lambda: _lib.call("bbdm_conv3x3_winograd_f32", x.data_ptr(), Cin, pw.data_ptr(), b.data_ptr(), None,
                                       0, o2.data_ptr(), Cout, 0, ws.data_ptr(), N, H, W, Cin, Cout, st), args.reps)
        err = float((o1 - o2).abs().max() / o1.abs().max())
        fl = 18.0 * N * H * W * Cout * Cin
        td += ms_d
        tw += ms_w
        print(f"N{N} {H}x{W} {Cin}->{Cout}: direct {ms_d:8.3f} ms ({fl / ms_d / 1e9:6.1f} TF)  winograd {ms_w:8.3f} ms "
              f"({fl / ms_w / 1e9:6.1f} TF-equivalent)  x{ms_d / ms_w:5.2f}  maxdiff {err:.2e}", flush=True)
        del x, w, o1, o2, ws
    print(f"total: direct {td:.2f} ms  winograd {tw:.2f} ms")


if __name__ == "__main__":
    main()
