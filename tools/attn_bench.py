#!/usr/bin/env python
"""The C2 attention (N 16, T 4096, 16 heads x 64) on bbdm_attention_f32, on the pre-split pair (K / V planes + attention) and on the pair's
fp16-plane form (round 6), HIP events.
    python tools/attn_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from bbdm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shape", default="16,4096,16,64")
    args = ap.parse_args()
    N, T, heads, ch = (int(v) for v in args.shape.split(","))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    C = heads * ch
    qkv = torch.randn(N, T, 3 * C, device=dev)
    out0, out1 = torch.empty(N, T, C, device=dev), torch.empty(N, T, C, device=dev)
    nb = lib.bbdm_attention_kv_planes_bytes(N, T, heads, ch)
    planes = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    fl = 4.0 * N * heads * T * T * ch

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
    one = lambda: _lib.call("bbdm_attention_f32", qkv.data_ptr(), 3 * C, out0.data_ptr(), C, None, N, T, heads, ch, 0, st)
    t0 = timed(one)
    print(f"one launch      : {t0:6.3f} ms  {fl / t0 / 1e9:6.1f} TF/s")
    if nb:
        kv = lambda: _lib.call("bbdm_attention_kv_planes_f32", qkv.data_ptr(), 3 * C, planes.data_ptr(), nb, N, T, heads, ch, 0, st)
        at = lambda: _lib.call("bbdm_attention_planes_f32", qkv.data_ptr(), 3 * C, out1.data_ptr(), C, None, N, T, heads, ch, 0, planes.data_ptr(), st)
        t1, t2 = timed(kv), timed(at)
        print(f"K / V planes    : {t1:6.3f} ms  ({(N * T * 2 * C * 10) / t1 / 1e6:6.1f} GB/s)")
        print(f"planes attention: {t2:6.3f} ms  {fl / t2 / 1e9:6.1f} TF/s   pair {t1 + t2:6.3f} ms   bit-equal {torch.equal(out0, out1)}")
        # ... on the fp16-pair planes (round 6): one scale from a bound of qkv -- the exact maximum, and 2^13 above it (where the planner's
        # provable bound of a qkv projection sits); error of all three forms against an fp64 attention of image 0, head 0
        nb2 = lib.bbdm_attention_kv_planes_h2_bytes(N, T, heads, ch)
        out2 = torch.empty(N, T, C, device=dev)
        q, k, v = (z[0, :, :ch].double() for z in (qkv[..., 0:ch], qkv[..., ch:2 * ch], qkv[..., 2 * ch:3 * ch]))   # (legacy order: head 0 = q | k | v)
        ref = torch.softmax((q @ k.t()) * ch ** -0.5, dim=-1) @ v
        err = lambda o: float((o[0, :, :ch].double() - ref).abs().max() / ref.abs().max())
        print(f"error vs fp64   : one launch {err(out0):.2e}, bf16x3 pair {err(out1):.2e}")
        for slack in (1.0, 8192.0):
            bound = (qkv.abs().max() * slack).reshape(1).float()
            kv2 = lambda: _lib.call("bbdm_attention_kv_planes_h2_f32", qkv.data_ptr(), 3 * C, planes.data_ptr(), nb2, N, T, heads, ch, 0, bound.data_ptr(), st)
            at2 = lambda: _lib.call("bbdm_attention_planes_h2_f32", qkv.data_ptr(), 3 * C, out2.data_ptr(), C, None, N, T, heads, ch, 0, planes.data_ptr(), bound.data_ptr(), st)
            t3 = timed(kv2)
            t4 = timed(at2)
            print(f"fp16-pair form, bound = {slack:g} x max |qkv|: K / V planes {t3:6.3f} ms, attention {t4:6.3f} ms  {fl / t4 / 1e9:6.1f} TF/s   "
                  f"pair {t3 + t4:6.3f} ms   error vs fp64 {err(out2):.2e}")


if __name__ == "__main__":
    main()
