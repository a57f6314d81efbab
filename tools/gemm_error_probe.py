#!/usr/bin/env python
"""Error of the plane GEMMs against fp64 as a function of the contraction length, next to a CPU model of the accumulation
(one round-to-nearest of the fp32 accumulator per MFMA): which share of the Winograd-domain error is the GEMM's chain of roundings.
Usage (GPU box): python tools/gemm_error_probe.py [--mode bf3|h2]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_ops as ops  # noqa: E402


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def fp16r(x):
    return x.to(torch.float16).to(torch.float32)


def chain(Vp, Up, terms, K, chunk=16):
    acc = torch.zeros(Vp[0].shape[0], Vp[0].shape[1], Up[0].shape[2], dtype=torch.float32)
    Vd, Ud = [v.double() for v in Vp], [u.double() for u in Up]
    for c in range(0, K, chunk):
        for ia, ib in terms:
            acc = (acc.double() + torch.bmm(Vd[ia][:, :, c:c + chunk], Ud[ib][:, c:c + chunk, :])).float()
    return acc


def split_bf3(x):
    x1 = bf16r(x)
    r = x - x1
    x2 = bf16r(r)
    return [x1, x2, bf16r(r - x2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf3")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, T, N = 8, 512, 128
    for K in (128, 256, 512, 1024, 2048, 4096):
        g = torch.Generator().manual_seed(K)
        V = torch.randn(B, T, K, generator=g)
        W = torch.randn(B, K, N, generator=g) * 0.05
        wp = W.reshape(B, K // 16, 16, N).permute(0, 1, 3, 2).contiguous()          # [batch][K/16][CoutPad][16]
        ref = torch.bmm(V.double(), W.double())
        rms = lambda m: float(((m.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
        mx = lambda m: float((m.double() - ref).abs().max() / ref.abs().max())
        if args.mode == "bf3":
            M = ops.gemm_bf3p(V.to(dev), wp.to(dev), B, K, N).cpu()
            sim = chain(split_bf3(V), split_bf3(W), [(1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0)], K)
        else:
            M = ops.gemm_h2p(V.to(dev), wp.to(dev), B, K, N).cpu()
            sv, sw = 2.0 ** 10, 2.0 ** 16
            v1 = fp16r(V * sv)
            w1 = fp16r(W * sw)
            sim = chain([v1, fp16r(V * sv - v1)], [w1, fp16r(W * sw - w1)], [(0, 1), (1, 0), (0, 0)], K) / (sv * sw)
        f32 = torch.bmm(V, W)
        print(f"K={K:5d}  {args.mode} kernel rms {rms(M):.2e} max {mx(M):.2e} | model (1 RNE per MFMA) rms {rms(sim):.2e} max {mx(sim):.2e}"
              f" | torch fp32 bmm (CPU) rms {rms(f32):.2e}", flush=True)


if __name__ == "__main__":
    main()
