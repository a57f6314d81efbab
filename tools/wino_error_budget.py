#!/usr/bin/env python
"""Where the rounding error of an F(8x8, 3x3) layer comes from, and what each arithmetic would make of it (CPU simulation, round 6).

One layer (Cin -> 32 channels, 32x32 pixels) through the ten-point transforms with every stage in fp32, the tile GEMMs accumulated the way
the matrix core does it (tools/microbench/mfma_round.hip: the fp32 accumulator is rounded once per 8 k and MFMA), on the data sets of the
round-5 verdict.  Rows:
  exact        the GEMMs in fp64 (only the transforms round): what is left when the accumulation error is taken away
  bf3          bf16x3 planes, six terms (rounds 2 - 5)
  bf3_2acc     ... with the five small terms in an accumulator of their own (not buildable: the 256 x 256 tile has no registers for it)
  f32mfma      v_mfma_f32_32x32x2_f32: one rounding per 2 k
  h2           fp16-pair planes, three terms (round 6), under a bound 2^9 above the data (what the GroupNorm bound is like)
  h2_2acc      ... main term and small terms in separate accumulators
  in64 / out64 fp64 arithmetic inside the 1-D passes of the input / output transform (round-5 verdict item 1c): no effect
Usage: python tools/wino_error_budget.py [Cin ...] [--points 5/4,9/4,2/5,4/5]  -> profiles/r06_wino_error_budget.txt"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_winograd_math_cpu import cook_toom  # noqa: E402


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def fp16r(x):
    return x.to(torch.float16).to(torch.float32)


def datasets(C, K=32, S=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    xg = F.silu(torch.randn(1, C, S, S, generator=g) * 1.5 + 0.3)
    wg = torch.randn(K, C, 3, 3, generator=g) * 0.02
    out = {"gauss": (xg, wg), "dc": (xg, wg + 0.05), "gain3": (xg, wg * 3)}
    torch.manual_seed(seed)
    out["student"] = (xg, torch.distributions.StudentT(3.0).sample((K, C, 3, 3)) * 0.02)
    xo = xg.clone()
    for i, j in torch.randint(0, S, (40, 2), generator=g).tolist():
        xo[:, :, i, j] *= 30
    out["outlier"] = (xo, wg)
    return out


def chain(Vp, Up, terms, K, per=8):
    """acc += sum over `per` k of V[ia] U[ib], rounded to fp32 -- once per `per` k and term (the matrix core: per = 8)."""
    acc = torch.zeros(Vp[0].shape[0], Vp[0].shape[1], Up[0].shape[2], dtype=torch.float32)
    Vd, Ud = [v.double() for v in Vp], [u.double() for u in Up]
    for c in range(0, K, 16):
        for ia, ib in terms:
            for h in range(0, 16, per):
                acc = (acc.double() + torch.bmm(Vd[ia][:, :, c + h:c + h + per], Ud[ib][:, c + h:c + h + per, :])).float()
    return acc


def layer(x, w, mats, mode, m=8):
    BT, G, AT = mats
    a = m + 2
    N, C, H, W = x.shape
    Kc = w.shape[0]
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)
    di = torch.float64 if mode == "in64" else torch.float32
    v = torch.einsum("ij,nctwjk->nctwik", BT.to(di), tiles.to(di))
    v = torch.einsum("nctwik,lk->nctwil", v, BT.to(di)).float()
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G)
    nt = v.shape[2] * v.shape[3]
    Vb = v.permute(4, 5, 0, 2, 3, 1).reshape(a * a, N * nt, C)
    Ub = U.permute(2, 3, 1, 0).reshape(a * a, C, Kc)
    six = [(1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0)]
    if mode in ("exact", "in64", "out64"):
        M = torch.bmm(Vb.double(), Ub).float() if mode == "exact" else None
    if mode in ("in64", "out64", "bf3", "bf3_2acc"):
        def split(t):
            t1 = bf16r(t)
            r = t - t1
            t2 = bf16r(r)
            return [t1, t2, bf16r(r - t2)]
        Vp, Up = split(Vb), split(Ub.float())
        if mode == "bf3_2acc":
            M = (chain(Vp, Up, [(0, 0)], C).double() + chain(Vp, Up, six[:5], C).double()).float()
        else:
            M = chain(Vp, Up, six, C)
    elif mode == "f32mfma":
        M = chain([Vb], [Ub.float()], [(0, 0)], C, per=2)
    elif mode in ("h2", "h2_2acc"):
        sv = 2.0 ** (math.floor(math.log2(2.0 ** 14 / float(Vb.abs().max()))) - 9)
        su = 2.0 ** math.floor(math.log2(2.0 ** 14 / float(Ub.abs().max())))
        v1, u1 = fp16r(Vb * sv), fp16r(Ub.float() * su)
        Vp, Up = [v1, fp16r(Vb * sv - v1)], [u1, fp16r(Ub.float() * su - u1)]
        if mode == "h2":
            M = chain(Vp, Up, [(0, 1), (1, 0), (0, 0)], C) / (sv * su)
        else:
            M = ((chain(Vp, Up, [(0, 0)], C).double() + chain(Vp, Up, [(0, 1), (1, 0)], C).double()) / (sv * su)).float()
    M = M.reshape(a, a, N, v.shape[2], v.shape[3], Kc).permute(2, 5, 3, 4, 0, 1)
    do = torch.float64 if mode == "out64" else torch.float32
    y = torch.einsum("ij,nktwjl->nktwil", AT.to(do), M.to(do))
    y = torch.einsum("nktwil,ml->nktwim", y, AT.to(do)).float()
    return y.permute(0, 1, 2, 4, 3, 5).reshape(N, Kc, H, W)


def main():
    torch.set_num_threads(8)
    pts = ["5/4", "9/4", "2/5", "4/5"]
    args = []
    for a in sys.argv[1:]:
        if a.startswith("--points"):
            pts = a.split("=")[1].split(",")
        else:
            args.append(int(a))
    print(f"F(8x8, 3x3) on the points 0, +-{', +-'.join(pts)}, inf: max-norm / rms error against the fp64 convolution")
    mats = cook_toom(pts, 8)
    for C in (args or [256, 1024]):
        ds = datasets(C)
        refs = {k: F.conv2d(x.double(), w.double(), padding=1) for k, (x, w) in ds.items()}
        for mode in ("exact", "bf3", "bf3_2acc", "f32mfma", "h2", "h2_2acc", "in64", "out64"):
            row = []
            for k, (x, w) in ds.items():
                d = layer(x, w, mats, mode).double() - refs[k]
                row.append(f"{k} {float(d.abs().max() / refs[k].abs().max()):.1e}/{float((d.pow(2).mean() / refs[k].pow(2).mean()).sqrt()):.1e}")
            print(f"Cin={C:5d} {mode:9s} " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
