#!/usr/bin/env python
"""How much do the F(4x4,3x3) layers of the latent models lose to launch quantisation?  36 transform points = 4.5 batch entries per XCD:
the same tile GEMM timed with 32 / 36 / 40 / 64 entries, and the 4 left-over entries as a split-K launch.
    python tools/quant_probe.py [--reps 30]"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib  # noqa: E402

SHAPES = [(512, 1024, 1024), (2048, 512, 512), (512, 2048, 1024), (2048, 1024, 1024), (2048, 1024, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for T, K, C in SHAPES:
        B = 64
        V = torch.randn(B, T, K, device=dev)
        W = torch.randn(B * K * C, device=dev) * 0.02
        ap_ = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(B, T, K), dtype=torch.uint8, device=dev)
        bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(B, K, C), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_split_rows_f32", V.data_ptr(), K, ap_.data_ptr(), B, T, K, st)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", W.data_ptr(), bp.data_ptr(), B, K, C, st)
        M = torch.empty(8 * 8 * T * C + B * T * C, device=dev)
        del V, W

        def timed(fn):
            for _ in range(3):
                fn()
            evs = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            return ts[len(ts) // 2]
        out = []
        for nb in (8, 16, 32, 36, 40, 64):
            t = timed(lambda: _lib.call("bbdm_gemm_bf3p_f32", ap_.data_ptr(), bp.data_ptr(), None, None, 0, M.data_ptr(), C, nb, T, K, C, st))
            out.append(f"b{nb} {t * 1e3:6.1f} us {2.0 * nb * T * K * C / t / 1e9:5.1f} TF")
        for nb, sp in ((4, 8), (4, 4), (8, 4)):
            if K // 16 // sp < 1:
                continue
            t = timed(lambda: _lib.call("bbdm_gemm_bf3p_splitk_f32", ap_.data_ptr(), bp.data_ptr(), M.data_ptr(), C, nb, T, T, K, C, sp, st))
            out.append(f"b{nb}/split{sp} {t * 1e3:6.1f} us")
        print(f"T{T} K{K} C{C}: " + " | ".join(out), flush=True)
        del ap_, bp, M
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
