#!/usr/bin/env python
"""1x1 weight gradients of the LBBDM-f4 training step (batch 32) on bbdm_conv_wgrad_f32's TN-GEMM path (gemm_tn_f32 + tn_finish4 +
colsum): ms and TFLOP/s per layer shape; BBDM_TN_TARGET / BBDM_TN_MINK (split-count knobs of tn_geom) are read by the library."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
import kernel_ops as ops  # noqa: E402

SHAPES = [(131072, 256, 128), (131072, 640, 128), (32768, 1024, 512), (32768, 128, 512), (32768, 1536, 512), (32768, 640, 512),
          (8192, 1024, 1024), (8192, 1024, 3072), (8192, 1536, 1024), (8192, 2048, 1024), (8192, 512, 1024)]
COUNT = {(131072, 256, 128): 2, (8192, 2048, 1024): 2}
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


total = 0.0
print(f"BBDM_TN_TARGET={os.environ.get('BBDM_TN_TARGET')} BBDM_TN_MINK={os.environ.get('BBDM_TN_MINK')}")
print("| pixels | Cin | Cout | ms | TFLOP/s |")
print("|---|---|---|---|---|")
for P, Ci, Co in SHAPES:
    x = torch.randn(P // 64, 8, 8, Ci, device=dev)
    dy = torch.randn(P // 64, 8, 8, Co, device=dev)
    ms = timed(lambda: ops.conv_wgrad(x, dy, Ci, Co, 1, with_bias=True))
    total += ms * COUNT.get((P, Ci, Co), 1)
    print(f"| {P} | {Ci} | {Co} | {ms:.3f} | {2.0 * P * Ci * Co / ms / 1e9:.1f} |")
print(f"sum over the step's 13 layers: {total:.3f} ms")
