#!/usr/bin/env python
"""Do the HBM-bound Winograd transforms overlap with the MFMA-bound tile GEMMs when they run on two HIP streams?

The tile GEMM workgroup (1024 threads x 123 VGPRs, 96 KB LDS) fills a CU's register file, so the overlap, if any, is CU-granular: the
dispatcher hands CUs to whichever queue has a workgroup ready.  The probe times, for a batch HALF (8 images) of the C2 layer shapes:
  gemm     : the 64 tile GEMMs of half A alone (stream 1), R times back to back
  transf   : input transform + output transform of half B alone (stream 2), R times
  both     : the two loops launched together on their streams, wall time until both are done
  serial   : the same launches interleaved on ONE stream (today's plan)
overlap = (serial - both) / min(gemm, transf): 1.0 = the shorter loop is hidden completely, 0 = no gain.

    python tools/overlap_probe.py [--reps 6]"""
import argparse
import os
import sys
import time

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib
import kernel_ops as ops  # noqa: E402

SHAPES = [  # N (half batch), H, W, Cin, Cout, launches per C2 step
    (8, 64, 64, 1024, 1024, 10),
    (8, 256, 256, 128, 128, 7),
    (8, 128, 128, 512, 512, 6),
    (8, 256, 256, 512, 512, 2),
    (8, 128, 128, 1024, 1024, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    R = args.reps
    dev = torch.device("cuda:0")
    lib = _lib.load()
    m, P = 6, 64
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    tot = [0.0, 0.0, 0.0, 0.0]
    for N, H, W, Cin, Cout, cnt in SHAPES:
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        st0 = torch.cuda.current_stream().cuda_stream
        x = torch.randn(N, H, W, Cin, device=dev)
        res = torch.randn(N, H, W, Cout, device=dev)
        out = torch.empty(N, H, W, Cout, device=dev)
        bias = torch.randn(Cout, device=dev)
        sc = torch.rand(N, Cin, device=dev) + 0.5
        bi = torch.randn(N, Cin, device=dev) * 0.1
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st0)
        VpA = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        VpB = torch.empty_like(VpA)
        MA = torch.empty(P * tiles * Cout, device=dev)
        MB = torch.randn(P * tiles * Cout, device=dev)
        _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, VpA.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W,
                  Cin, st0)
        torch.cuda.synchronize()

        def gemm(st):
            _lib.call("bbdm_winograd_gemm_bf3p_f32", m, VpA.data_ptr(), Bp.data_ptr(), MA.data_ptr(), N, H, W, Cin, Cout, st)

        def transf(st):
            _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, VpB.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0,
                      N, H, W, Cin, st)
            _lib.call("bbdm_winograd_output_f32", m, MB.data_ptr(), bias.data_ptr(), res.data_ptr(), Cout, out.data_ptr(), Cout, 0,
                      N, H, W, Cout, st)

        def wall(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / R

        def only_gemm():
            for _ in range(R):
                gemm(s1.cuda_stream)

        def only_transf():
            for _ in range(R):
                transf(s2.cuda_stream)

        def both():
            for _ in range(R):
                gemm(s1.cuda_stream)
                transf(s2.cuda_stream)

        def serial():
            for _ in range(R):
                gemm(s1.cuda_stream)
                transf(s1.cuda_stream)

        tg, tt, tb, ts = wall(only_gemm), wall(only_transf), wall(both), wall(serial)
        for i, v in enumerate((tg, tt, tb, ts)):
            tot[i] += cnt * v
        print(f"N{N} {H}x{W} {Cin}->{Cout} x{cnt}: gemm {tg:6.3f}  transf {tt:6.3f}  serial {ts:6.3f}  both {tb:6.3f} ms  "
              f"overlap {(ts - tb) / min(tg, tt):5.2f}", flush=True)
        del x, res, out, pw, Bp, VpA, VpB, MA, MB
    print(f"weighted (half batch): gemm {tot[0]:.2f} transf {tot[1]:.2f} serial {tot[3]:.2f} both {tot[2]:.2f} ms")


if __name__ == "__main__":
    main()
