#!/usr/bin/env python
"""Do the HBM-bound Winograd transforms of one half of a batch overlap with the MFMA-bound tile GEMMs of the other half when the two
run on streams that own DISJOINT compute units (bbdm_stream_create_partition: hipExtStreamCreateWithCUMask)?

tools/overlap_probe.py showed that two plain streams do not overlap (the tile GEMM's workgroups fill every CU's register file, the
dispatcher runs the two queues back to back).  Here the streaming launches get the CU-mask bits [0, t) -- t / 8 CUs of every XCD --
and the matrix launches the other 256 - t CUs; the persistent tile GEMM sizes its grid for its stream's share.  Per C2 layer shape, for
a batch HALF (8 images), R repetitions each:
  serial  : input transform, tile GEMMs, output transform back to back on ONE ordinary stream (today's plan), per repetition
  gemm@G  : the tile GEMMs alone on the G partition            transf@T : the two transforms alone on the T partition
  both    : the two loops at the same time on their partitions, wall time per repetition (= what a two-chain schedule pays per layer
            and half when neither chain waits for the other)
The C2-weighted sums say what a static partition can win: serial - both.

    python tools/partition_probe.py [--reps 6] [--t 32,64,96]"""
import argparse
import ctypes
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
    (8, 64, 64, 2048, 1024, 2),
    (8, 256, 256, 512, 512, 2),
    (8, 128, 128, 1024, 1024, 2),
    (8, 256, 256, 640, 128, 1),
    (8, 128, 128, 1536, 512, 1),
]


def partition(lib, lo, hi):
    h = ctypes.c_void_p()
    _lib.check(lib.bbdm_stream_create_partition(lo, hi, ctypes.byref(h)), "bbdm_stream_create_partition")
    return torch.cuda.ExternalStream(h.value), h.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--t", default="32,64,96")
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--coreside", action="store_true",
                    help="no CU masks: two plain streams, the tile GEMM as ONE 8-wave workgroup per CU (256 x 128 tiles, LDS padded), the "
                         "transforms co-resident on the same CUs")
    args = ap.parse_args()
    R = args.reps
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cus = lib.bbdm_device_cus()
    m, P = 6, 64
    ts = [int(v) for v in args.t.split(",")]
    parts = {}
    if args.coreside:
        import ctypes as _ct
        lib.bbdm_debug_set_bf3p_one_per_cu.restype = _ct.c_int
        lib.bbdm_debug_set_bf3p_kernel.restype = _ct.c_int
        ts = [0]
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        parts[0] = (sa, sa.cuda_stream, sb, sb.cuda_stream)
    for t in ([] if args.coreside else ts):
        sT, hT = partition(lib, 0, t)
        sG, hG = partition(lib, t, cus)
        parts[t] = (sT, hT, sG, hG)
        print(f"# partition t={t}: T stream {lib.bbdm_stream_cus(hT)} CUs, G stream {lib.bbdm_stream_cus(hG)} CUs", flush=True)
    s0 = torch.cuda.Stream()
    shapes = SHAPES if args.shapes is None else [SHAPES[int(i)] for i in args.shapes.split(",")]
    tot = {t: [0.0, 0.0, 0.0, 0.0] for t in ts}
    tot_serial = 0.0
    for N, H, W, Cin, Cout, cnt in shapes:
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

        def gemm(st, one=False):
            if one:                 # 256 x 128 pipe kernel, one workgroup per CU
                lib.bbdm_debug_set_bf3p_kernel(5)
                lib.bbdm_debug_set_bf3p_one_per_cu(1)
            _lib.call("bbdm_winograd_gemm_bf3p_f32", m, VpA.data_ptr(), Bp.data_ptr(), MA.data_ptr(), N, H, W, Cin, Cout, st)
            if one:
                lib.bbdm_debug_set_bf3p_kernel(6)
                lib.bbdm_debug_set_bf3p_one_per_cu(0)

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

        def serial():
            for _ in range(R):
                transf(s0.cuda_stream)
                gemm(s0.cuda_stream)

        tser = wall(serial)
        tot_serial += cnt * tser
        line = f"N{N} {H}x{W} {Cin}->{Cout} x{cnt}: serial {tser:6.3f} |"
        for t in ts:
            sT, hT, sG, hG = parts[t]

            def only_gemm():
                for _ in range(R):
                    gemm(hG, args.coreside)

            def only_transf():
                for _ in range(R):
                    transf(hT)

            def both():
                for _ in range(R):
                    gemm(hG, args.coreside)
                    transf(hT)

            tg, tt, tb = wall(only_gemm), wall(only_transf), wall(both)
            for i, v in enumerate((tg, tt, tb, max(tg, tt))):
                tot[t][i] += cnt * v
            line += f" t={t}: gemm@G {tg:6.3f} transf@T {tt:6.3f} both {tb:6.3f} |"
        print(line, flush=True)
        del x, res, out, pw, Bp, VpA, VpB, MA, MB
    print(f"C2-weighted (half batch, x2 = per step): serial {tot_serial:.2f} ms")
    for t in ts:
        a = tot[t]
        print(f"  t={t:3d}: gemm@G {a[0]:.2f}  transf@T {a[1]:.2f}  both {a[2]:.2f}  (max of the two alone {a[3]:.2f})  "
              f"gain per step {2 * (tot_serial - a[2]):.2f} ms")


if __name__ == "__main__":
    main()
