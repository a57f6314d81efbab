#!/usr/bin/env python
"""Diagnostic: per-parameter gradient error of the full-size (237 M) UNet at a 64x64 latent, batch 2, in module order,
against autograd on the oracle -- to locate where in the backward chain an error starts.  python tools/diag_c4_grads.py"""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests"), os.path.join(_ROOT, "oracle")]
import test_fullsize_parity_gpu as T  # noqa: E402

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
up = dict(T.UNET_PIXEL, image_size=S, in_channels=3, condition_key="nocond")
m, sd = T._model(up, T.BB, 4040, dev)
m.train()
g = torch.Generator().manual_seed(77)
N = 2
x0 = torch.randn(N, 3, S, S, generator=g)
y = torch.randn(N, 3, S, S, generator=g)
t = torch.tensor([812, 37])
nz = torch.randn(N, 3, S, S, generator=g)
l32, g32 = T._oracle_loss_grads(sd, up, T.BB, x0, y, t, nz, torch.float32)
gmax = max(float(v.abs().max()) for v in g32.values())
for wino in (0, 6):
    m.denoise_fn.winograd = wino
    m.denoise_fn._plans = {}
    m.zero_grad(set_to_none=True)
    loss, _ = m.p_losses(x0.to(dev), y.to(dev), None, t.to(dev), nz.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    print(f"== winograd={wino} S={S} loss {float(loss.detach()):.7f} oracle {l32:.7f}")
    for k, p in m.denoise_fn.named_parameters():
        ref = g32[k]
        own = float(ref.abs().max())
        e_own = float((p.grad.cpu() - ref).abs().max()) / max(own, 1e-30)
        cos = float(torch.nn.functional.cosine_similarity(p.grad.cpu().flatten().double(), ref.flatten().double(), dim=0))
        ratio = float(p.grad.cpu().double().norm() / ref.double().norm().clamp_min(1e-30))
        print(f"{k:55s} |g|max {own:.2e} ({own / gmax:.1e} of gmax)  err/own {e_own:.2e}  cos {cos:.6f}  norm ratio {ratio:.5f}")
