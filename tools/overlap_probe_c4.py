#!/usr/bin/env python
"""Do two INDEPENDENT tile GEMMs of the LBBDM-f4 training step (the data-gradient and weight-gradient GEMMs of one layer) fill each
other's last rounds when they run on two HIP streams?  (End of round 5: the mid-size GEMMs execute 1.56 / 3.125 rounds as 2 / 4 --
DESIGN.md 5.)  Times, per shape: each GEMM alone, both on one stream, both on two streams.

    python tools/overlap_probe_c4.py [--reps 8]"""
import argparse
import os
import sys
import time

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib

SHAPES = [  # m, N, H, W, Cin, Cout
    (8, 32, 32, 32, 512, 512),
    (8, 32, 32, 32, 1024, 1024),
    (8, 32, 32, 32, 1536, 512),
    (4, 32, 16, 16, 1024, 1024),
    (4, 32, 16, 16, 2048, 1024),
    (8, 32, 64, 64, 128, 128),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    R = args.reps
    dev = torch.device("cuda:0")
    lib = _lib.load()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for m, N, H, W, Cin, Cout in SHAPES:
        P = (m + 2) ** 2
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        u8 = lambda n: torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
        # forward-type GEMM (Cin -> Cout) and the data-gradient GEMM (Cout -> Cin) on planes of random bytes (bf16 bit patterns)
        Va, Ba = u8(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin)), u8(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout))
        Vb, Bb = u8(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cout)), u8(lib.bbdm_gemm_bf3p_b_bytes(P, Cout, Cin))
        Ma, Mb = torch.empty(P * tiles * Cout, device=dev), torch.empty(P * tiles * Cin, device=dev)
        # the weight-gradient GEMM: dU = V^T dM over the tiles
        At, Bt = u8(lib.bbdm_gemm_bf3p_tn_at_bytes(P, tiles, Cin)), u8(lib.bbdm_gemm_bf3p_tn_bt_bytes(P, tiles, Cout))
        dU = torch.empty(lib.bbdm_gemm_bf3p_tn_splits(P, tiles, Cin, Cout) * P * Cin * Cout, device=dev)
        for t in (Va, Ba, Vb, Bb, At, Bt):      # (keep the exponent fields sane: clear the top exponent bit of every bf16)
            t.view(torch.int16).bitwise_and_(0x3FFF)
        torch.cuda.synchronize()

        def g_fwd(st):
            _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Va.data_ptr(), Ba.data_ptr(), Ma.data_ptr(), N, H, W, Cin, Cout, st)

        def g_dgrad(st):
            _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vb.data_ptr(), Bb.data_ptr(), Mb.data_ptr(), N, H, W, Cout, Cin, st)

        def g_wgrad(st):
            _lib.call("bbdm_gemm_bf3p_tn_f32", At.data_ptr(), Bt.data_ptr(), dU.data_ptr(), P, tiles, Cin, Cout, st)

        def wall(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / R

        def loop(*pairs):
            def run():
                for _ in range(R):
                    for f, s in pairs:
                        f(s.cuda_stream)
            return run

        td, tw = wall(loop((g_dgrad, s1))), wall(loop((g_wgrad, s1)))
        ser = wall(loop((g_dgrad, s1), (g_wgrad, s1)))
        two = wall(loop((g_dgrad, s1), (g_wgrad, s2)))
        print(f"F{m} N{N} {H}x{W} {Cin}->{Cout}: dgrad {td:6.3f}  wgrad {tw:6.3f}  one stream {ser:6.3f}  two streams {two:6.3f} ms  "
              f"({(1 - two / ser) * 100:+.1f} %)", flush=True)
        del Va, Ba, Vb, Bb, Ma, Mb, At, Bt, dU


if __name__ == "__main__":
    main()
