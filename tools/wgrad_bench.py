#!/usr/bin/env python
"""Micro-benchmark of bbdm_conv_wgrad_f32 on the LBBDM-f4 training shapes (batch 32)."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]      # kernel_ops lives with the tests
import kernel_ops as ops  # noqa: E402

SHAPES = [(32, 16, 16, 1024, 1024, 3), (32, 32, 32, 512, 512, 3), (32, 64, 64, 128, 128, 3), (32, 16, 16, 2048, 1024, 3),
          (32, 64, 64, 512, 512, 3), (32, 32, 32, 1024, 512, 1)]
dev = torch.device("cuda:0")
tot_ms = tot_fl = 0.0
for N, H, W, Ci, Co, ks in SHAPES:
    x = torch.randn(N, H, W, Ci, device=dev)
    dy = torch.randn(N, H, W, Co, device=dev)
    ops.conv_wgrad(x, dy, Ci, Co, ks, with_bias=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv_wgrad(x, dy, Ci, Co, ks, with_bias=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * N * H * W * Co * Ci * ks * ks
    tot_ms += ms
    tot_fl += fl
    print(f"N{N} {H}x{W} {Ci}->{Co} k{ks}: {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF")
print(f"total {tot_ms:.3f} ms {tot_fl / tot_ms / 1e9:.1f} TF")
