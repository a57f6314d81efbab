#!/usr/bin/env python
"""Direct f32 implicit GEMM vs Winograd F(m x m, 3x3) on the bf16x3 pipe GEMM (input planes -> gemm_bf3p -> output transform) on the
SMALL 3x3 layers: the latent configurations (c3 / c5), the 64^2 pixel model (c1) and the first stage -- the layers `winograd_tile`
(bbdm_amd/unet.py) sent to the direct kernel on round-2 measurements of the f32-MFMA tile GEMM.

    python tools/small_conv_bench.py [--reps 10]"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib
from bbdm_amd.unet import winograd_tile
import kernel_ops as ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout
    (32, 16, 16, 128, 128),       # c5 (LBBDM-f16 latent 16x16, batch 32)
    (32, 16, 16, 128, 512),
    (32, 8, 8, 512, 512),
    (32, 8, 8, 1024, 512),
    (32, 8, 8, 512, 1024),
    (32, 4, 4, 1024, 1024),
    (32, 4, 4, 2048, 1024),
    (32, 8, 8, 2048, 1024),
    (32, 8, 8, 1536, 512),
    (32, 16, 16, 1024, 512),
    (32, 16, 16, 640, 128),
    (32, 16, 16, 256, 128),
    (4, 64, 64, 128, 128),        # c1 (64^2 pixels, batch 4)
    (4, 64, 64, 256, 128),
    (4, 32, 32, 512, 512),
    (4, 32, 32, 1024, 512),
    (4, 16, 16, 1024, 1024),
    (4, 16, 16, 2048, 1024),
    (4, 16, 16, 512, 1024),
    (32, 32, 32, 512, 512),       # c3 / c4 (LBBDM-f4 latent 64x64, batch 32)
    (32, 16, 16, 1024, 1024),
    (32, 16, 16, 2048, 1024),
    (32, 64, 64, 128, 128),
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
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for N, H, W, Cin, Cout in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
        b = torch.randn(Cout, device=dev)
        pd = ops.pack_conv_weight(w)
        o1 = torch.empty(N, H, W, Cout, device=dev)
        o2 = torch.empty(N, H, W, Cout, device=dev)
        ms_d = _time(lambda: ops.conv2d_nhwc(x, pd, b, Cout, 3, out=o1), args.reps)
        fl = 18.0 * N * H * W * Cout * Cin
        line = f"N{N} {H}x{W} {Cin}->{Cout} (plan: m={winograd_tile(N, H, W, Cin, Cout)}): direct {ms_d:7.3f} ms ({fl / ms_d / 1e9:6.1f} TF)"
        for m in (2, 4, 6):
            if m != 6 and (H % m or W % m):
                continue
            P = (m + 2) ** 2
            tiles = lib.bbdm_winograd_tiles(m, N, H, W)
            if not lib.bbdm_gemm_bf3p_supported(tiles, Cin, Cout):
                continue
            pw = ops.pack_winograd_weight(w, m=m)
            Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
            _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
            Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
            ks0 = lib.bbdm_winograd_gemm_bf3p_splits(m, N, H, W, Cin, Cout)
            t_in = _time(lambda: _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, Vp.data_ptr(), None, None, 0, 0, 0,
                                           N, H, W, Cin, st), args.reps)
            raw = N * -(-H // m) * -(-W // m)
            line += f"\n    F{m} ({raw}/{tiles} tiles, lib splits {ks0}): input {t_in:.3f} |"
            best = None
            for ks in sorted({1, 2, 4, 8, ks0}):
                if m == 6 and ks > 1 or Cin // 16 < ks * 2:
                    continue
                M = torch.empty(ks * P * tiles * Cout, device=dev)
                try:
                    t_g = _time(lambda: _lib.call("bbdm_winograd_gemm_bf3p_splitk_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(),
                                                  N, H, W, Cin, Cout, ks, st), args.reps)
                except Exception as e:      # a split count that leaves an empty split
                    continue
                t_o = _time(lambda: _lib.call("bbdm_winograd_output_splitk_stats_f32", m, M.data_ptr(), b.data_ptr(), None, 0,
                                              o2.data_ptr(), Cout, 0, N, H, W, Cout, None, 0, 0, None, 0, 0, ks, st), args.reps)
                err = float((o1 - o2).abs().max() / o1.abs().max())
                tot = t_in + t_g + t_o
                line += f" k{ks}{'*' if ks == ks0 else ''}: {t_g:.3f}+{t_o:.3f}={tot:.3f} x{ms_d / tot:4.2f} ({err:.0e}) |"
                del M
            del pw, Bp, Vp
        print(line, flush=True)
        del x, w, o1, o2


if __name__ == "__main__":
    main()
