#!/usr/bin/env python
"""Does running a Winograd layer over CHUNKS of the batch keep its V / M intermediates in the 256 MiB Infinity Cache?

Images are independent through the whole UNet (GroupNorm statistics are per image), so a plan may run input transform -> tile GEMMs
-> output transform over n images at a time, reusing ONE chunk-sized V / M workspace that then stays on die.  This probe runs
the three stages of the layer shapes of the C2 step for a batch of 16 in chunks of n = 16 (today), 8, 4, 2, 1 and prints ms per
stage summed over the chunks (HIP events on the launch stream between the stages) and the V + M footprint of a chunk.

    python tools/mall_probe.py [--reps 5] [--shapes 0,2]"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib
import kernel_ops as ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, launches per C2 step
    (16, 64, 64, 1024, 1024, 10),
    (16, 256, 256, 128, 128, 7),
    (16, 128, 128, 512, 512, 6),
    (16, 64, 64, 2048, 1024, 2),
    (16, 256, 256, 512, 512, 2),
    (16, 128, 128, 1024, 1024, 2),
    (16, 256, 256, 640, 128, 1),
    (16, 128, 128, 1536, 512, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--chunks", default="16,8,4,2,1")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    m, P = 6, 64
    shapes = SHAPES if args.shapes is None else [SHAPES[int(i)] for i in args.shapes.split(",")]
    chunks = [int(c) for c in args.chunks.split(",")]
    tot = {n: [0.0, 0.0, 0.0] for n in chunks}
    for N, H, W, Cin, Cout, cnt in shapes:
        x = torch.randn(N, H, W, Cin, device=dev)
        res = torch.randn(N, H, W, Cout, device=dev)
        out = torch.empty(N, H, W, Cout, device=dev)
        bias = torch.randn(Cout, device=dev)
        sc = torch.rand(N, Cin, device=dev) + 0.5
        bi = torch.randn(N, Cin, device=dev) * 0.1
        pw = ops.pack_winograd_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02, m=m)
        Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
        ref = None
        for n in chunks:
            tiles = lib.bbdm_winograd_tiles(m, n, H, W)
            vbytes = lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin)
            Vp = torch.empty(vbytes, dtype=torch.uint8, device=dev)
            M = torch.empty(P * tiles * Cout, device=dev)
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N // n)]

            def run(record):
                for c in range(N // n):
                    xs, rs, os_ = x[c * n:], res[c * n:], out[c * n:]
                    if record:
                        ev[c][0].record()
                    _lib.call("bbdm_winograd_input_bf3p_f32", m, xs.data_ptr(), Cin, Vp.data_ptr(), sc[c * n:].data_ptr(),
                              bi[c * n:].data_ptr(), Cin, 1, 0, n, H, W, Cin, st)
                    if record:
                        ev[c][1].record()
                    _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), n, H, W, Cin, Cout, st)
                    if record:
                        ev[c][2].record()
                    _lib.call("bbdm_winograd_output_f32", m, M.data_ptr(), bias.data_ptr(), rs.data_ptr(), Cout, os_.data_ptr(), Cout,
                              0, n, H, W, Cout, st)
                    if record:
                        ev[c][3].record()

            run(False)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = torch.equal(ref, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run(False)
            e1.record()
            torch.cuda.synchronize()
            total = e0.elapsed_time(e1) / args.reps
            acc = [0.0, 0.0, 0.0]
            for _ in range(args.reps):
                run(True)
                torch.cuda.synchronize()
                for c in range(N // n):
                    for s in range(3):
                        acc[s] += ev[c][s].elapsed_time(ev[c][s + 1]) / args.reps
            for s in range(3):
                tot[n][s] += cnt * acc[s]
            print(f"N{N} {H}x{W} {Cin}->{Cout} x{cnt} chunk {n:2d}: V+M {(vbytes + M.numel() * 4) / 2**20:7.1f} MiB | input {acc[0]:6.3f} "
                  f"gemm {acc[1]:6.3f} output {acc[2]:6.3f} ms | total {total:6.3f} ms {'==' if same else '!= MISMATCH'}", flush=True)
            del Vp, M
        del x, res, out, pw, Bp
    for n in chunks:
        a = tot[n]
        print(f"C2-weighted chunk {n:2d}: input {a[0]:.2f} gemm {a[1]:.2f} output {a[2]:.2f} sum {sum(a):.2f} ms")


if __name__ == "__main__":
    main()
