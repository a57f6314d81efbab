#!/usr/bin/env python
"""The wide 1x1 convolutions of the C2 step on bbdm_conv1x1_bf3_f32 (8-wave workgroups, A split while staged) against
bbdm_conv1x1_bf3q_f32 (the pipelined kernel: every wave splits its share of the next-but-one chunk between its MFMAs).

    python tools/conv1x1_bench.py [--reps 8]"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib

SHAPES = [  # pixels, Cin, Cout, residual, launches per C2 step
    (65536, 1024, 3072, 0, 1), (65536, 1024, 1024, 1, 1), (65536, 2048, 1024, 0, 2), (65536, 1536, 1024, 0, 1),
    (262144, 1536, 512, 0, 1), (262144, 1024, 512, 0, 1), (262144, 640, 512, 0, 1), (1048576, 640, 128, 0, 1),
    (1048576, 256, 128, 0, 2), (262144, 128, 512, 0, 1), (65536, 512, 1024, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    tot = [0.0, 0.0]
    for pixels, cin, cout, res, cnt in SHAPES:
        x = torch.randn(pixels, cin, device=dev)
        w = torch.randn(cout, cin, device=dev) * 0.03
        b = torch.randn(cout, device=dev)
        r = torch.randn(pixels, cout, device=dev) if res else None
        pf = torch.empty(lib.bbdm_conv_packed_floats(cout, cin, 1), dtype=torch.float32, device=dev)
        _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), pf.data_ptr(), cout, cin, cin, 1, st)
        pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(1, cin, cout), dtype=torch.int16, device=dev)
        _lib.call("bbdm_gemm_bf3_pack_f32", pf.data_ptr(), pk.data_ptr(), 1, cin, cout, st)
        bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(1, cin, cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pf.data_ptr(), bp.data_ptr(), 1, cin, cout, st)
        o0 = torch.empty(pixels, cout, device=dev)
        o1 = torch.empty(pixels, cout, device=dev)
        rp = None if r is None else r.data_ptr()

        def old():
            _lib.call("bbdm_conv1x1_bf3_f32", x.data_ptr(), cin, pk.data_ptr(), b.data_ptr(), rp, cout if res else 0, o0.data_ptr(), cout,
                      pixels, cin, cout, st)

        def new():
            _lib.call("bbdm_conv1x1_bf3q_f32", x.data_ptr(), cin, bp.data_ptr(), b.data_ptr(), rp, cout if res else 0, o1.data_ptr(), cout,
                      pixels, cin, cout, st)

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.reps

        t0, t1 = timed(old), timed(new)
        fl = 2.0 * pixels * cin * cout
        tot[0] += cnt * t0
        tot[1] += cnt * t1
        print(f"pixels {pixels:8d} {cin:5d}->{cout:5d} res{res} x{cnt}: bf3 {t0:6.3f} ms ({fl / t0 / 1e9:5.1f} TF)  bf3q {t1:6.3f} ms ({fl / t1 / 1e9:5.1f} TF)  "
              f"{'bit-equal' if torch.equal(o0, o1) else 'MISMATCH'}", flush=True)
        del x, w, o0, o1, pf, pk, bp
    print(f"C2-weighted: bf3 {tot[0]:.2f} ms  bf3q {tot[1]:.2f} ms")


if __name__ == "__main__":
    main()
