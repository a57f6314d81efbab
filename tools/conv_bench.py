#!/usr/bin/env python
"""Micro-benchmark of bbdm_conv2d_nhwc_f32 on the layer shapes that dominate the 256^2 / batch-16 step.

    python tools/conv_bench.py [--reps 5]
Prints ms and TFLOP/s per shape (HIP events on the launch stream)."""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]      # kernel_ops lives with the tests
import kernel_ops as ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, ks
    (16, 64, 64, 1024, 1024, 3),
    (16, 128, 128, 512, 512, 3),
    (16, 256, 256, 128, 128, 3),
    (16, 64, 64, 2048, 1024, 3),
    (16, 128, 128, 1024, 512, 1),
    (16, 256, 256, 256, 128, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tot_ms = tot_fl = 0.0
    for N, H, W, Cin, Cout, ks in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.02
        b = torch.randn(Cout, device=dev)
        pw = ops.pack_conv_weight(w)
        out = torch.empty(N, H, W, Cout, device=dev)
        ops.conv2d_nhwc(x, pw, b, Cout, ks, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.conv2d_nhwc(x, pw, b, Cout, ks, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        fl = 2.0 * N * H * W * Cout * Cin * ks * ks
        tot_ms += ms
        tot_fl += fl
        print(f"N{N} {H}x{W} {Cin}->{Cout} k{ks}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
    print(f"total: {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
