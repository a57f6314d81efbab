#!/usr/bin/env python
"""The batched tile-GEMM stage of the Winograd path alone (bbdm_winograd_gemm_f32 = conv_igemm_f32 in 1x1 mode).

    python tools/gemm_bench.py [--reps 5] [--m 4]
Prints ms and executed TFLOP/s per shape (HIP events on the launch stream)."""
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
    (16, 128, 128, 512, 512),
    (16, 128, 128, 1024, 1024),
    (16, 256, 256, 512, 512),
    (16, 256, 256, 128, 128),
    (32, 32, 32, 512, 512),
    (32, 16, 16, 1024, 1024),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--m", type=int, default=6)
    ap.add_argument("--bf3", type=int, default=1, help="1 = bf16x3 kernel (csrc/gemm_bf3.hip), 0 = f32 MFMA kernel")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    m, P = args.m, (args.m + 2) ** 2
    tot_ms = tot_fl = 0.0
    for N, H, W, Cin, Cout in SHAPES:
        tiles = lib.bbdm_winograd_tiles(m, N, H, W)
        V = torch.randn(P * tiles * Cin, device=dev)
        M = torch.empty(P * tiles * Cout, device=dev)
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        if args.bf3:
            pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(P, Cin, Cout), dtype=torch.int16, device=dev)
            _lib.call("bbdm_gemm_bf3_pack_f32", pw.data_ptr(), pk.data_ptr(), P, Cin, Cout, st)
            call = lambda: _lib.call("bbdm_winograd_gemm_bf3_f32", m, V.data_ptr(), pk.data_ptr(), M.data_ptr(), N, H, W, Cin,
                                     Cout, st)
        else:
            call = lambda: _lib.call("bbdm_winograd_gemm_f32", m, V.data_ptr(), pw.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout,
                                     st)
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        fl = 2.0 * P * N * (H // m) * (W // m) * Cin * Cout
        tot_ms += ms
        tot_fl += fl
        print(f"F{m} N{N} {H}x{W} {Cin}->{Cout}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
        del V, M, pw
    print(f"total: {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
