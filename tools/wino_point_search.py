#!/usr/bin/env python
"""Search for the F(8x8, 3x3) point set {0, +-p1, +-p2, +-p3, +-p4, inf} with the smallest error UNDER THE ACCUMULATION THE MATRIX CORE
PERFORMS (fp16-pair planes, three terms, one rounding of the accumulator per 8 k and MFMA: tools/wino_error_budget.py's `h2` row): all
23 751 four-pair subsets of 29 rationals whose B^T / A^T are exact in fp32, scored on Gaussian, Student-t and 30x-outlier data (Cin = 256).
    python tools/wino_point_search.py PART NPARTS      (writes /tmp/w/ptsearch_PART.json; four parts of ~27 min each in round 6)
Result (profiles/r06_point_search.txt): {5/4, 9/4, 2/5, 4/5} -- rms 22 - 26 % below round 5's {1/2, 3/4, 4/3, 2} at Cin = 256 and 1024."""
import sys, itertools, math, random, torch, time, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch.nn.functional as F
from fractions import Fraction as Fr
from test_winograd_math_cpu import cook_toom
torch.set_num_threads(1)

def fp16r(x): return x.to(torch.float16).to(torch.float32)

def make_data(C=256, K=16, S=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = F.silu(torch.randn(1, C, S, S, generator=g) * 1.5 + 0.3)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.02
    torch.manual_seed(seed)
    ws = torch.distributions.StudentT(3.0).sample((K, C, 3, 3)) * 0.02
    xo = x.clone()
    for i, j in torch.randint(0, S, (20, 2), generator=g).tolist(): xo[:, :, i, j] *= 30
    return [(x, w), (x, ws), (xo, w)]

def wino_h2(x, w, mats, m=8):
    BT, G, AT = mats
    a = m + 2
    N, C, H, W = x.shape; Kc = w.shape[0]
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)
    bt = BT.float()
    v = torch.einsum("ij,nctwjk->nctwik", bt, tiles)
    v = torch.einsum("nctwik,lk->nctwil", v, bt)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G).float()
    nt = v.shape[2] * v.shape[3]
    Vb = v.permute(4, 5, 0, 2, 3, 1).reshape(a * a, N * nt, C)
    Ub = U.permute(2, 3, 1, 0).reshape(a * a, C, Kc)
    sv = 2.0 ** math.floor(math.log2(2.0 ** 14 / float(Vb.abs().max()) / 16)); su = 2.0 ** math.floor(math.log2(2.0 ** 14 / float(Ub.abs().max())))
    v1 = fp16r(Vb * sv); v2 = fp16r(Vb * sv - v1); u1 = fp16r(Ub * su); u2 = fp16r(Ub * su - u1)
    acc = torch.zeros(a * a, N * nt, Kc, dtype=torch.float32)
    Vd = [v1.double(), v2.double()]; Ud = [u1.double(), u2.double()]
    # hardware: one rounding of the accumulator per 8 k and MFMA; per 16-k chunk the terms come (0,1) (1,0) (0,0)
    for c in range(0, C, 16):
        for (ia, ib) in ((0, 1), (1, 0), (0, 0)):
            for h in (0, 8):
                acc = (acc.double() + torch.bmm(Vd[ia][:, :, c + h:c + h + 8], Ud[ib][:, c + h:c + h + 8, :])).float()
    M = (acc / (sv * su)).reshape(a, a, N, v.shape[2], v.shape[3], Kc).permute(2, 5, 3, 4, 0, 1)
    at = AT.float()
    y = torch.einsum("ij,nktwjl->nktwil", at, M)
    y = torch.einsum("nktwil,ml->nktwim", y, at)
    return y.permute(0, 1, 2, 4, 3, 5).reshape(N, Kc, H, W)

def score(pts, data, refs):
    mats = cook_toom(pts, 8)
    BT, G, AT = mats
    if not (torch.equal(BT.float().double(), BT) and torch.equal(AT.float().double(), AT)): return None
    out = []
    for (x, w), ref in zip(data, refs):
        d = wino_h2(x, w, mats).double() - ref
        out.append((float(d.abs().max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())))
    return out

if __name__ == '__main__':
    part, nparts = int(sys.argv[1]), int(sys.argv[2])
    pool = ["1/4","3/8","1/2","5/8","2/3","3/4","5/6","7/8","1","9/8","6/5","5/4","4/3","3/2","5/3","7/4","2","9/4","5/2","3","4","1/3","2/5","3/5","4/5","7/6","8/5","7/2","3/16"]
    data = make_data()
    refs = [F.conv2d(x.double(), w.double(), padding=1) for x, w in data]
    cands = list(itertools.combinations(pool, 4))
    random.Random(1).shuffle(cands)
    cands = [("1/2","3/4","4/3","2")] + cands
    res = []
    t0 = time.time()
    for i, pts in enumerate(cands[part::nparts]):
        # prune: the product of |p| spread too large -> hopeless
        vals = [float(Fr(p)) for p in pts]
        if max(vals) / min(vals) > 12: continue
        s = score(list(pts), data, refs)
        if s is None: continue
        res.append((pts, s))
        if len(res) % 50 == 0:
            res.sort(key=lambda r: sum(e[1] for e in r[1]))
            os.makedirs('/tmp/w', exist_ok=True); json.dump(res[:40], open(f'/tmp/w/ptsearch_{part}.json', 'w'))
            print(part, i, len(res), time.time() - t0, res[0], flush=True)
        if time.time() - t0 > 3000: break
    res.sort(key=lambda r: sum(e[1] for e in r[1]))
    os.makedirs('/tmp/w', exist_ok=True); json.dump(res[:40], open(f'/tmp/w/ptsearch_{part}.json', 'w'))
    print("done", part, len(res))
