"""The Winograd F(m x m, 3x3) algebra used by csrc/winograd.hip, restated with explicit matrices (CPU, no GPU):
B^T, G, A^T below are the matrices the HIP kernels hard-code as adds / multiplies (bt_transform, g_transform,
at_transform; m = 6 is the experimental 8x8-tile path).  Checks that they reproduce a 3x3 'same' convolution exactly in fp64, that the data-gradient variant
(transposed + flipped filter) is the transposed convolution, and the fp32 rounding levels DESIGN.md §4.5 quotes."""
import pytest
import torch
import torch.nn.functional as F

MATS = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]], dtype=torch.float64)),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                      [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1.]], dtype=torch.float64),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]],
                     dtype=torch.float64)),
    # m = 6 (experimental path): the 8x8 transform with points {0, +-1, +-2, +-1/2, inf}
    6: (torch.tensor([[1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0],
                      [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0], [0, 1 / 2, 1 / 4, -5 / 2, -5 / 4, 2, 1, 0],
                      [0, -1 / 2, 1 / 4, 5 / 2, -5 / 4, -2, 1, 0], [0, 2, 4, -5 / 2, -5, 1 / 2, 1, 0],
                      [0, -2, 4, 5 / 2, -5, -1 / 2, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1.]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45],
                      [1 / 90, -1 / 45, 2 / 45], [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1.]],
                     dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0],
                      [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 0], [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0],
                      [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1.]], dtype=torch.float64)),
}


def cook_toom(pairs, m, r=3):
    """Cook-Toom matrices (B^T, G, A^T) of F(m, r) on the points {0, +-p for p in pairs, inf}, in exact rational arithmetic:
    A^T[i][j] = p_j^i, G[j][k] = p_j^k / prod_{l != j} (p_j - p_l), B^T row j = the coefficients of prod_{l != j} (x - p_l) (row inf: of
    prod_l (x - p_l)) -- then every row of B^T and column of A^T is multiplied by the odd part of its common denominator and a power
    of two (largest entry in [1, 2)), G taking the reciprocals: B^T and A^T become dyadic rationals, i.e. exact in fp32, as
    csrc/winograd_math.h hard-codes them for m = 8."""
    from fractions import Fraction as Fr
    import math
    pts = [Fr(0)] + [s * Fr(p) for p in pairs for s in (1, -1)]
    n = m + r - 1
    assert len(pts) == n - 1

    def polymul(a, b):
        c = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                c[i + j] += x * y
        return c
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * r for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    for j, p in enumerate(pts):
        for i in range(m):
            AT[i][j] = p ** i
        Nj = math.prod([p - q for k, q in enumerate(pts) if k != j], start=Fr(1))
        for k in range(r):
            G[j][k] = p ** k / Nj
        f = [Fr(1)]
        for k, q in enumerate(pts):
            if k != j:
                f = polymul(f, [-q, Fr(1)])
        BT[j][:len(f)] = f
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][r - 1] = Fr(1)
    f = [Fr(1)]
    for q in pts:
        f = polymul(f, [-q, Fr(1)])
    BT[n - 1][:len(f)] = f

    def odd_den(vals):
        den = 1
        for c in vals:
            d = c.denominator
            while d % 2 == 0:
                d //= 2
            den = den * d // math.gcd(den, d)
        return Fr(den)

    def pow2_norm(vals):
        mx, e = max(abs(c) for c in vals), 0
        while mx >= 2:
            mx, e = mx / 2, e - 1
        while mx < 1:
            mx, e = mx * 2, e + 1
        return Fr(2) ** e
    for j in range(n):
        s = odd_den(BT[j])
        s *= pow2_norm([c * s for c in BT[j]])
        BT[j] = [c * s for c in BT[j]]
        G[j] = [c / s for c in G[j]]
        col = [AT[i][j] for i in range(m)]
        s = odd_den(col)
        s *= pow2_norm([c * s for c in col])
        for i in range(m):
            AT[i][j] *= s
        G[j] = [c / s for c in G[j]]
    t64 = lambda M_: torch.tensor([[float(c) for c in row] for row in M_], dtype=torch.float64)
    return t64(BT), t64(G), t64(AT)


# m = 8: ten points {0, +-5/4, +-9/4, +-2/5, +-4/5, inf} (round 6; round 5: {+-1/2, +-3/4, +-4/3, +-2}); B^T and A^T exact in fp32 (checked
# below); csrc/winograd_math.h's m = 8 branches are GENERATED from this set by tools/gen_winograd8.py
POINTS8 = ["5/4", "9/4", "2/5", "4/5"]
MATS[8] = cook_toom(POINTS8, 8)


def test_f8_matrices_are_exact_in_fp32_and_the_point_set_is_the_accurate_one():
    """B^T / A^T of m = 8 survive the round trip through fp32 (the algebra then holds exactly; only data round), and the point set
    beats the textbook {0, +-1, +-2, +-1/2, +-4, inf} by an order of magnitude in fp32 (every stage of the restatement in fp32, Cin = 256,
    post-SiLU-like data: rms 4.6e-5 against 4.6e-4; m = 6: 6.2e-6 -- the price of the larger tile is 7x m = 6's error)."""
    BT, G, AT = MATS[8]
    assert torch.equal(BT.float().double(), BT) and torch.equal(AT.float().double(), AT)
    g = torch.Generator().manual_seed(8)
    x = F.silu(torch.randn(1, 256, 24, 24, generator=g) * 1.5 + 0.3)
    w = torch.randn(8, 256, 3, 3, generator=g) * 0.03
    ref = F.conv2d(x.double(), w.double(), padding=1)
    rms = lambda y: float(((y.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    e_ours = rms(winograd_conv(x, w, 8, torch.float32))
    MATS[80] = cook_toom(["1", "2", "1/2", "4"], 8)
    try:
        e_textbook = rms(_winograd_conv_mats(x, w, 8, MATS[80], torch.float32))
    finally:
        del MATS[80]
    e6 = rms(winograd_conv(x, w, 6, torch.float32))
    MATS[81] = cook_toom(["1/2", "3/4", "4/3", "2"], 8)
    try:
        e_round5 = rms(_winograd_conv_mats(x, w, 8, MATS[81], torch.float32))
    finally:
        del MATS[81]
    print("fp32 restatement, rms: this set", e_ours, "round 5's", e_round5, "textbook", e_textbook, "m = 6", e6)
    assert e_ours < 1e-4 and e_textbook > 4 * e_ours and e6 < e_ours and e_ours < 1.05 * e_round5, (e_ours, e_round5, e_textbook, e6)


def test_f8_constants_are_the_generated_ones_and_the_gain_bounds_the_transform():
    """tools/gen_winograd8.py regenerates the m = 8 branches of csrc/winograd_math.h from POINTS8 without changing the file, and
    wino_input_gain(m) (the factor the fp16-pair planes' bound is multiplied by, csrc/h2_split.h) is >= max_i (sum_j |B^T_ij|)^2 and
    within 1 % of it, for every tile."""
    import importlib.util
    import os
    from bbdm_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_winograd8", os.path.join(root, "tools", "gen_winograd8.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    blocks = gen.generate(POINTS8)
    src = open(os.path.join(root, "bbdm_amd", "csrc", "winograd_math.h")).read()
    for key in ("bt", "at", "g", "a", "gt"):
        assert blocks[key] in src, key
    lib = _lib.load()
    for m in (2, 4, 6, 8):
        BT, G, AT = MATS[m]
        for name, fn, exact in (("input", lib.bbdm_winograd_input_gain, float(BT.abs().sum(1).max()) ** 2),
                                ("g", lib.bbdm_winograd_g_gain, float(G.abs().sum(1).max()) ** 2),
                                ("dy", lib.bbdm_winograd_dy_gain, float(AT.abs().sum(0).max()) ** 2)):
            got = float(fn(m))
            assert exact <= got <= 1.01 * exact, (m, name, exact, got)
    assert float(lib.bbdm_winograd_input_gain(7)) == float(lib.bbdm_winograd_input_gain(6))
    g72 = float(MATS72[1].abs().sum(1).max()) ** 2
    assert g72 <= float(lib.bbdm_winograd_g_gain(7)) <= 1.01 * g72


# F(7x7, 2x2) on the same eight points (round 5: the phase filters of conv3x3(nearest x2 (x)) read 2 x 2 pixels each): B^T is m = 6's,
# G is 8 x 2 with the same row scalings, A^T gains the x^6 row and moves the point-at-infinity column there.
_PTS = [0, 1, -1, 2, -2, .5, -.5]
_F = [1, -2 / 9, -2 / 9, 1 / 90, 1 / 90, 32 / 45, 32 / 45]
MATS72 = (MATS[6][0],
          torch.tensor([[_F[j], _F[j] * _PTS[j]] for j in range(7)] + [[0, 1.]], dtype=torch.float64),
          torch.tensor([[_PTS[j] ** i for j in range(7)] + [1. if i == 6 else 0.] for i in range(7)], dtype=torch.float64))


def test_f72_exact_in_fp64_and_as_upsample_phases():
    """y[a][b] = sum_{r,s < 2} g[r][s] d[a + r][b + s] for a, b < 7 from one 8 x 8 window; and the four phase filters of
    conv3x3(nearest x2 (x)) evaluated this way -- phase (pa, pb) of tile (th, tw) lands at x-grid rows 7 th + a - pa, columns
    7 tw + b - pb (csrc/winograd_math.h: wino_tdim) -- reproduce the convolution of the upsampled image."""
    BT, G, AT = MATS72
    g = torch.Generator().manual_seed(72)
    D = torch.randn(8, 8, generator=g, dtype=torch.float64)
    K = torch.randn(2, 2, generator=g, dtype=torch.float64)
    Y = AT @ ((G @ K @ G.T) * (BT @ D @ BT.T)) @ AT.T
    ref = torch.stack([torch.stack([(K * D[a:a + 2, b:b + 2]).sum() for b in range(7)]) for a in range(7)])
    assert float((Y - ref).abs().max()) < 1e-12
    N, C, Ko, H, W = 2, 3, 4, 9, 15
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Ko, C, 3, 3, generator=g, dtype=torch.float64)
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    out = torch.full((N, Ko, 2 * H, 2 * W), float("nan"), dtype=torch.float64)
    TH, TW = (H + 7) // 7, (W + 7) // 7
    xp = F.pad(x, (1, 7 * TW + 7 - W, 1, 7 * TH + 7 - H))                        # window rows 7 t - 1 .. 7 t + 6 -> padded index 7 t ..
    for pa in range(2):
        rows = [w[:, :, 0], w[:, :, 1] + w[:, :, 2]] if pa == 0 else [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]     # [r][K, C, kx]
        for pb in range(2):
            g2 = torch.stack([torch.stack([r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if pb == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]], -1)
                              for r in rows], -2)                              # [K, C, 2, 2]
            U = torch.einsum("ij,kcjl,ml->kcim", G, g2, G)
            for th in range(TH):
                for tw in range(TW):
                    d = xp[:, :, 7 * th:7 * th + 8, 7 * tw:7 * tw + 8]
                    V = torch.einsum("ij,ncjk,lk->ncil", BT, d, BT)
                    Mx = torch.einsum("ncil,kcil->nkil", V, U)
                    Yt = torch.einsum("ij,nkjl,ml->nkim", AT, Mx, AT)
                    for a in range(7):
                        for b in range(7):
                            oh, ow = 7 * th + a - pa, 7 * tw + b - pb
                            if 0 <= oh < H and 0 <= ow < W:
                                out[:, :, 2 * oh + pa, 2 * ow + pb] = Yt[:, :, a, b]
    assert not bool(torch.isnan(out).any())
    assert float((out - want).abs().max()) < 1e-11


def winograd_conv(x, w, m, dtype):
    """x [N,C,H,W], w [K,C,3,3] -> [N,K,H,W]; every stage in `dtype` (the HIP path: fp32; m = 8: the filter transform in fp64, rounded
    once, as csrc/winograd.hip evaluates it)."""
    return _winograd_conv_mats(x, w, m, MATS[m], dtype)


def _winograd_conv_mats(x, w, m, mats, dtype):
    BT, G, AT = (t.to(dtype) for t in mats)
    if m == 8 and dtype == torch.float32:
        G = mats[1]
    x, w = x.to(dtype), w.to(G.dtype)
    a = m + 2
    N, C, H, W = x.shape
    K = w.shape[0]
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)            # N, C, th, tw, a, a
    V = torch.einsum("ij,nctwjk,lk->nctwil", BT, tiles, BT)                  # input transform  B^T d B
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G).to(dtype)                  # filter transform G g G^T
    M = torch.einsum("nctwil,kcil->nktwil", V, U)                            # (m+2)^2 independent GEMMs over c
    Y = torch.einsum("ij,nktwjl,ml->nktwim", AT, M, AT)                      # output transform A^T m A
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


@pytest.mark.parametrize("m", [2, 4, 6, 8])
def test_exact_in_fp64(m):
    g = torch.Generator().manual_seed(m)
    x = torch.randn(2, 5, 24 if m == 8 else 12, 24, generator=g, dtype=torch.float64)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, padding=1)
    tol = 1e-11 if m == 8 else 1e-12             # (fp64 rounding amplified by the ten-point transforms: 1.5e-12 measured)
    assert float((winograd_conv(x, w, m, torch.float64) - ref).abs().max()) < tol
    # data gradient = the same algorithm on the transposed, spatially flipped filter (winograd_weight_kernel, dgrad)
    dy = torch.randn(2, 7, 24 if m == 8 else 12, 24, generator=g, dtype=torch.float64)
    wd = w.transpose(0, 1).flip(2, 3)
    assert float((winograd_conv(dy, wd, m, torch.float64) - F.conv_transpose2d(dy, w, padding=1)).abs().max()) < tol


def winograd_wgrad(x, dy, m, dtype):
    """dW [K,C,3,3] of y = conv(x, w) from x [N,C,H,W] and dy [N,K,H,W] in the Winograd domain (csrc/winograd_wgrad.hip):
    dU_xi = sum_tiles V_xi^T dM_xi with V = B^T d B (the forward's input transform), dM = A dY A^T; dW = G^T dU G."""
    BT, G, AT = (t.to(dtype) for t in MATS[m])
    if m == 8 and dtype == torch.float32:
        G = MATS[m][1]                     # (m = 8: G^T dU G in fp64, rounded once -- wgrad_finish_*_kernel<8>)
    x, dy = x.to(dtype), dy.to(dtype)
    a = m + 2
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)            # N, C, th, tw, a, a
    V = torch.einsum("ij,nctwjk,lk->nctwil", BT, tiles, BT)
    dyt = dy.unfold(2, m, m).unfold(3, m, m)                                  # N, K, th, tw, m, m
    dM = torch.einsum("ji,nktwjl,lm->nktwim", AT, dyt, AT)                    # A dY A^T
    dU = torch.einsum("nctwil,nktwil->kcil", V, dM)                           # (m+2)^2 GEMMs over the tiles
    return torch.einsum("ij,kcil,lm->kcjm", G, dU.to(G.dtype), G).to(dtype)   # G^T dU G


@pytest.mark.parametrize("m", [2, 4, 6, 8])
def test_wgrad_exact_in_fp64(m):
    g = torch.Generator().manual_seed(10 + m)
    H = 24 if m == 8 else 12
    x = torch.randn(2, 5, H, 24, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(2, 7, H, 24, generator=g, dtype=torch.float64)
    F.conv2d(x, w, padding=1).backward(dy)
    assert float((winograd_wgrad(x.detach(), dy, m, torch.float64) - w.grad).abs().max()) < (1e-10 if m == 8 else 1e-11)


def test_wgrad_fp32_rounding_levels():
    """Weight gradient of a 60x60 layer (K = 2 x 3600 pixels per weight), every stage in fp32: rms error relative
    to the rms of the fp64 gradient.  Measured: direct (torch fp32) 5.3e-7, m = 2: 5.6e-7, m = 4: 2.8e-6, m = 6: 5.3e-6."""
    g = torch.Generator().manual_seed(1)
    x = F.silu(torch.randn(2, 32, 60, 60, generator=g))
    dy = torch.randn(2, 16, 60, 60, generator=g) * 1e-3
    xr, wr = x.double().requires_grad_(), torch.zeros(16, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, padding=1).backward(dy.double())
    ref = wr.grad
    rms = lambda d: float(((d.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    e2, e4, e6 = (rms(winograd_wgrad(x, dy, m, torch.float32)) for m in (2, 4, 6))
    assert e2 < 2e-6 and e4 < 5e-6 and e6 < 1e-5, (e2, e4, e6)
    # m = 8 (64x64 here: whole tiles): the ten-point transform's rounding, as on the forward side several times m = 6's
    x8 = F.silu(torch.randn(2, 32, 64, 64, generator=g))
    dy8 = torch.randn(2, 16, 64, 64, generator=g) * 1e-3
    xr, wr = x8.double().requires_grad_(), torch.zeros(16, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, padding=1).backward(dy8.double())
    ref = wr.grad
    e8 = rms(winograd_wgrad(x8, dy8, 8, torch.float32))
    print("wgrad fp32 rounding, m = 8:", e8)
    assert e6 < e8 < 2e-4, (e6, e8)


def test_fp32_rounding_levels():
    """rms error relative to the rms of the fp64 result, Cin = 512, unit-variance post-SiLU-like activations."""
    g = torch.Generator().manual_seed(0)
    x = F.silu(torch.randn(1, 512, 24, 24, generator=g))
    w = torch.randn(64, 512, 3, 3, generator=g) * 0.02
    ref = F.conv2d(x.double(), w.double(), padding=1)
    rms = lambda y: float(((y.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    e_direct = rms(F.conv2d(x, w, padding=1))
    e2, e4, e6 = (rms(winograd_conv(x, w, m, torch.float32)) for m in (2, 4, 6))
    assert e_direct < 1e-6 and e2 < 2e-6 and e4 < 1e-5 and e6 < 2e-5          # measured: 2e-7, 5e-7, 3e-6, 6e-6
    assert e2 < e4 < e6                            # each step up in tile size costs accuracy: ~10x, then ~2x


@pytest.mark.parametrize("m", [2, 4, 6, 7, 8])
def test_hip_source_transforms_match_the_matrices(m):
    """csrc/winograd.hip evaluates B^T, A^T and G as hand-factored adds / multiplies; the same template code compiled
    for the host (bbdm_debug_winograd_transform_1d, an exported test hook) must agree with the matrices above on random
    vectors and on every unit vector (= every matrix column)."""
    import ctypes
    import numpy as np
    from bbdm_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    fn = lib.bbdm_debug_winograd_transform_1d
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    BT, G, AT = (t.numpy() for t in (MATS72 if m == 7 else MATS[m]))
    rng = np.random.RandomState(m)
    sides = ((0, BT), (1, AT), (2, G)) if m == 7 else ((0, BT), (1, AT), (2, G), (3, AT.T), (4, G.T))
    for which, mat in sides:      # 3, 4: the weight-gradient side (winograd_math.h)
        rows, cols = mat.shape
        vecs = [rng.randn(cols).astype(np.float32) for _ in range(8)] + list(np.eye(cols, dtype=np.float32))
        for v in vecs:
            v = np.ascontiguousarray(v)
            out = np.zeros(rows, dtype=np.float32)
            assert fn(m, which, v.ctypes.data, out.ctypes.data) == 0
            want = mat @ v.astype(np.float64)
            assert np.abs(out - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), (m, which, v, out, want)
