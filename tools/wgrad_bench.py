#!/usr/bin/env python
"""Weight-gradient micro-benchmark on the LBBDM-f4 training shapes (batch 32): the direct kernel (bbdm_conv_wgrad_f32) against
the Winograd-domain path (bbdm_conv3x3_winograd_wgrad_f32, m = 4 / 6), with the stages of the latter timed separately.
TFLOP/s are DIRECT-convolution FLOPs / time for every column (so the columns compare)."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]      # kernel_ops lives with the tests
import kernel_ops as ops  # noqa: E402
from bbdm_amd import _lib  # noqa: E402

SHAPES = [(32, 64, 64, 128, 128), (32, 64, 64, 256, 128), (32, 64, 64, 640, 128), (32, 64, 64, 512, 512),
          (32, 32, 32, 128, 512), (32, 32, 32, 512, 512), (32, 32, 32, 1024, 512), (32, 32, 32, 1536, 512),
          (32, 32, 32, 1024, 1024), (32, 16, 16, 512, 1024), (32, 16, 16, 1024, 1024), (32, 16, 16, 2048, 1024),
          (32, 16, 16, 1536, 1024)]
dev = torch.device("cuda:0")
lib = _lib.load()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = {"direct": 0.0, "best": 0.0}
print("| shape | direct ms (TF) | m=4 ms (TF) [in / dy / gemm / finish] | m=6 ms (TF) [in / dy / gemm / finish] |")
print("|---|---|---|---|")
for N, H, W, Ci, Co in SHAPES:
    x = torch.randn(N, H, W, Ci, device=dev)
    dy = torch.randn(N, H, W, Co, device=dev)
    fl = 2.0 * N * H * W * Co * Ci * 9
    st = torch.cuda.current_stream().cuda_stream
    ms_d = timed(lambda: ops.conv_wgrad(x, dy, Ci, Co, 3, with_bias=True))
    cells = [f"{ms_d:.3f} ({fl / ms_d / 1e9:.0f})"]
    best = ms_d
    for m in (4, 6):
        if m == 4 and (H % 4 or W % 4):
            cells.append("-")
            continue
        ws = torch.empty(lib.bbdm_winograd_wgrad_workspace_floats(m, N, H, W, Ci, Co), dtype=torch.float32, device=dev)
        dw = torch.empty(Co, Ci, 3, 3, device=dev)
        db = torch.empty(Co, device=dev)
        ms = timed(lambda: _lib.call("bbdm_conv3x3_winograd_wgrad_f32", m, x.data_ptr(), Ci, dy.data_ptr(), Co, dw.data_ptr(),
                                      db.data_ptr(), ws.data_ptr(), N, H, W, Ci, Co, st))
        P, Tp = (m + 2) ** 2, lib.bbdm_winograd_tiles(m, N, H, W)
        T = N * -(-H // m) * -(-W // m)
        V, dM = ws, ws[P * Tp * Ci:]
        dU = ws[P * Tp * (Ci + Co):]
        sp = lib.bbdm_gemm_tn_splits(P, T, Ci, Co)
        t_in = timed(lambda: _lib.call("bbdm_winograd_input_f32", m, x.data_ptr(), Ci, V.data_ptr(), None, None, 0, 0, 0, N, H, W, Ci, st))
        t_dy = timed(lambda: _lib.call("bbdm_winograd_dy_transform_f32", m, dy.data_ptr(), Co, dM.data_ptr(), N, H, W, Co, st))
        t_g = timed(lambda: _lib.call("bbdm_gemm_tn_batched_f32", V.data_ptr(), Ci, Tp * Ci, dM.data_ptr(), Co, Tp * Co, dU.data_ptr(),
                                      P, T, Ci, Co, st))
        t_f = timed(lambda: _lib.call("bbdm_winograd_wgrad_finish_f32", m, dU.data_ptr(), sp, dw.data_ptr(), Ci, Co, st))
        gtf = 2.0 * P * T * Ci * Co / t_g / 1e9
        cells.append(f"{ms:.3f} ({fl / ms / 1e9:.0f}) [{t_in:.3f} / {t_dy:.3f} / {t_g:.3f} = {gtf:.0f} TF x{sp} / {t_f:.3f}]")
        best = min(best, ms)
        del ws
    tot["direct"] += ms_d
    tot["best"] += best
    print(f"| N{N} {H}x{W} {Ci}->{Co} | " + " | ".join(cells) + " |", flush=True)
print(f"\ntotal direct {tot['direct']:.2f} ms, best-of {tot['best']:.2f} ms")
