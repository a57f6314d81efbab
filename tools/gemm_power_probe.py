#!/usr/bin/env python
"""Is the tile GEMM (csrc/gemm_bf3p.hip) bound by its own stalls or by what the part lets an MFMA stream draw?

For a few big layers of the C2 step the SAME launch is timed (HIP events around every launch)
  (a) back to back on real (random) operands,
  (b) back to back on ALL-ZERO planes (no toggling in the multiplier arrays: tools/microbench/mfma_peak.hip runs 1.3-1.8x faster on
      zeros -- a kernel that is limited by its own stalls does not care),
  (c) on real operands with an HBM-bound copy of ~0.5 ms between two launches (the step's duty cycle: the bare MFMA stream gains
      36 % from such gaps, profiles/r05_mfma_peak.txt),
  (d) zeros + gaps.
    python tools/gemm_power_probe.py [--reps 30]
"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib  # noqa: E402
import kernel_ops as ops  # noqa: E402

SHAPES = [(16, 64, 64, 1024, 1024), (16, 128, 128, 512, 512), (16, 64, 64, 2048, 1024), (16, 256, 256, 512, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    m, P = 6, 64
    src = torch.empty(300 << 20, dtype=torch.float32, device=dev).normal_()          # 1.2 GB: ~0.5 ms of copy
    dst = torch.empty_like(src)
    for N, H, W, Cin, Cout in SHAPES:
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        x = torch.randn(N, H, W, Cin, device=dev)
        Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        M = torch.empty(P * tiles * Cout, device=dev)
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
        _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, Vp.data_ptr(), None, None, 0, 0, 0, N, H, W, Cin, st)
        gemm = lambda: _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
        fl = 2.0 * P * tiles * Cin * Cout

        def run(gap):
            for _ in range(5):
                gemm()
            evs = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gemm()
                e1.record()
                evs.append((e0, e1))
                if gap:
                    dst.copy_(src)
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            return ts[len(ts) // 2]
        r = {}
        with _lib.option("bf3p_pad_rows", 0):
            r["real b2b, idle blocks re-read live rows"] = run(False)
        r["real b2b"], r["real gaps"] = run(False), run(True)
        Vp.zero_()
        Bp.zero_()
        r["zero b2b"], r["zero gaps"] = run(False), run(True)
        print(f"N{N} {H}x{W} {Cin}->{Cout}: " + " | ".join(f"{k} {v:6.3f} ms {fl / v / 1e9:6.1f} TF/s" for k, v in r.items()), flush=True)
        del x, Vp, M, Bp, pw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
