#!/usr/bin/env python
"""bf16x3 vs fp16-pair planes on single Winograd layers: (a) error against an fp64 convolution on the stress sets of the round-5 verdict
(Gaussian / DC filters / 3x gain / Student-t / spatial outliers), (b) time of the input transform and the tile GEMMs at the C2 / C3 layer
shapes.  GPU box:  python tools/h2_probe.py [--no-time] [--no-err]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_ops as ops  # noqa: E402
from bbdm_amd import _lib  # noqa: E402


def stress_sets(C, K, S, seed=0):
    g = torch.Generator().manual_seed(seed)
    xg = F.silu(torch.randn(2, C, S, S, generator=g) * 1.5 + 0.3)
    wg = torch.randn(K, C, 3, 3, generator=g) * 0.02
    out = {"gauss": (xg, wg), "dc": (xg, wg + 0.05), "gain3": (xg, wg * 3)}
    torch.manual_seed(seed)
    out["student"] = (xg, torch.distributions.StudentT(3.0).sample((K, C, 3, 3)) * 0.02)
    xo = xg.clone()
    for i, j in torch.randint(0, S, (40, 2), generator=g).tolist():
        xo[:, :, i, j] *= 30
    out["outlier"] = (xo, wg)
    return out


def errors(dev):
    print("== error vs fp64 conv: max|d|/max|ref| / rms ratio ==")
    for m in (8, 6):
        for C in (128, 512, 1024):
            S = 32 if m == 8 else 36
            for name, (x, w) in stress_sets(C, 128, S).items():
                ref = F.conv2d(x.double(), w.double(), padding=1)
                xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
                row = []
                for mode in ("bf3", "h2"):
                    o = ops.conv3x3_winograd_planes(xg, w.to(dev), None, m, mode=mode).permute(0, 3, 1, 2).cpu().double()
                    d = o - ref
                    row.append(f"{mode}: {float(d.abs().max() / ref.abs().max()):.2e} / {float((d.pow(2).mean() / ref.pow(2).mean()).sqrt()):.2e}")
                print(f"m={m} Cin={C:4d} {name:8s} " + "   ".join(row), flush=True)


def timing(dev):
    print("== time per launch (ms), bf16x3 planes vs fp16-pair planes ==")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(16, 64, 64, 1024, 1024), (16, 128, 128, 512, 512), (16, 128, 128, 256, 256), (16, 256, 256, 128, 128), (16, 64, 64, 2048, 1024),
              (16, 256, 256, 256, 128), (32, 64, 64, 128, 128), (32, 32, 32, 256, 256), (32, 32, 32, 512, 512), (32, 64, 64, 256, 128)]
    for (N, H, W, Cin, Cout) in shapes:
        m = 8
        planes, tiles = 100, lib.bbdm_winograd_tiles(m, N, H, W)
        x = torch.randn(N, H, W, Cin, device=dev)
        sc = torch.rand(N, Cin, device=dev) + 0.5
        bi = torch.randn(N, Cin, device=dev) * 0.1
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
        pf = torch.empty(lib.bbdm_winograd_packed_floats(m, Cout, Cin), dtype=torch.float32, device=dev)
        _lib.call("bbdm_winograd_pack_weight_f32", m, w.data_ptr(), pf.data_ptr(), Cout, Cin, Cin, 0, st)
        Tp = (tiles + 255) // 256 * 256
        M = torch.empty(planes * Tp * Cout, dtype=torch.float32, device=dev)
        ub = ops.absmax(w)
        vb = torch.full((1,), 40.0, dtype=torch.float32, device=dev)
        b3 = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(planes, Cin, Cout), dtype=torch.uint8, device=dev)
        bh = torch.empty(lib.bbdm_gemm_h2p_b_bytes(planes, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pf.data_ptr(), b3.data_ptr(), planes, Cin, Cout, st)
        _lib.call("bbdm_winograd_pack_weight_h2p_f32", m, w.data_ptr(), bh.data_ptr(), Cout, Cin, Cin, 0, ub.data_ptr(), st)
        V3 = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(planes, tiles, Cin), dtype=torch.uint8, device=dev)
        Vh = torch.empty(lib.bbdm_gemm_h2p_a_bytes(planes, tiles, Cin), dtype=torch.uint8, device=dev)
        calls = {
            "in  bf3": lambda: _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, V3.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W, Cin, st),
            "in  h2 ": lambda: _lib.call("bbdm_winograd_input_h2p_f32", m, x.data_ptr(), Cin, Vh.data_ptr(), sc.data_ptr(), bi.data_ptr(), Cin, 1, 0, N, H, W, Cin, vb.data_ptr(), st),
            "gemm bf3": lambda: _lib.call("bbdm_winograd_gemm_bf3p_f32", m, V3.data_ptr(), b3.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st),
            "gemm h2 ": lambda: _lib.call("bbdm_winograd_gemm_h2p_f32", m, Vh.data_ptr(), bh.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, vb.data_ptr(), ub.data_ptr(), st),
        }
        res = {}
        for k, fn in calls.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[k] = e0.elapsed_time(e1) / reps
        fl = 2.0 * planes * tiles * Cin * Cout
        print(f"N={N} {H}x{W} {Cin}->{Cout}: " + "  ".join(f"{k} {v:.3f}" for k, v in res.items()) +
              f"  | gemm TF/s fp32-eq bf3 {fl / res['gemm bf3'] / 1e9:.0f} h2 {fl / res['gemm h2 '] / 1e9:.0f}", flush=True)
        del x, V3, Vh, M, b3, bh, pf


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--no-err", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if not a.no_err:
        errors(dev)
    if not a.no_time:
        timing(dev)
