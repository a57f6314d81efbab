#!/usr/bin/env python
"""Direct implicit-GEMM 3x3 vs Winograd F(2x2,3x3) / F(4x4,3x3) on the wide layers of the 256^2 / batch-16 step.

    python tools/wino_bench.py [--reps 5]
Prints ms per shape for both paths (HIP events on the launch stream) and the max relative difference."""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]      # kernel_ops lives with the tests
from bbdm_amd import _lib
import kernel_ops as ops  # noqa: E402

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
    (32, 64, 64, 128, 128),       # c3 / c4 (LBBDM-f4 latent, batch 32)
    (32, 32, 32, 512, 512),
    (32, 32, 32, 1024, 512),
    (32, 16, 16, 1024, 1024),
    (32, 16, 16, 2048, 1024),
    (4, 32, 32, 512, 512),        # c1 (64^2 pixels, batch 4)
    (4, 16, 16, 1024, 1024),
    (32, 8, 8, 512, 512),         # c5 (LBBDM-f16 latent)
    (32, 4, 4, 1024, 1024),
]


MS = (2, 4, 6)


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
        pd = ops.pack_conv_weight(w)
        o1 = torch.empty(N, H, W, Cout, device=dev)
        o2 = torch.empty(N, H, W, Cout, device=dev)
        ms_d = _time(lambda: ops.conv2d_nhwc(x, pd, b, Cout, 3, out=o1), args.reps)
        fl = 18.0 * N * H * W * Cout * Cin
        td += ms_d
        line = f"N{N} {H}x{W} {Cin}->{Cout}: direct {ms_d:8.3f} ms ({fl / ms_d / 1e9:6.1f} TF)"
        best = ms_d
        for m in MS:
            if m != 6 and (H % m or W % m):
                continue
            pw = ops.pack_winograd_weight(w, m=m)
            ws = torch.empty(lib.bbdm_winograd_workspace_floats(m, N, H, W, Cin, Cout), device=dev)
            ms_w = _time(lambda: _lib.call("bbdm_conv3x3_winograd_f32", m, x.data_ptr(), Cin, pw.data_ptr(), b.data_ptr(),
                                           None, 0, o2.data_ptr(), Cout, 0, ws.data_ptr(), N, H, W, Cin, Cout, st), args.reps)
            err = float((o1 - o2).abs().max() / o1.abs().max())
            line += f" | F{m}: {ms_w:8.3f} ms x{ms_d / ms_w:5.2f} maxdiff {err:.1e}"
            best = min(best, ms_w)
            del ws, pw
        tw += best
        print(line, flush=True)
        del x, w, o1, o2
    print(f"total: direct {td:.2f} ms  best-of(direct, F2, F4[, F6]) {tw:.2f} ms")


if __name__ == "__main__":
    main()
