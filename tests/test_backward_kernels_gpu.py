"""Backward kernels (training path) against torch.autograd of the same op in fp32 on the CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

from fixtures import rel_err

pytestmark = pytest.mark.gpu
TOL = 3e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_BWD = [(2, 16, 16, 32, 64, 3), (1, 24, 40, 64, 128, 3), (3, 8, 8, 128, 96, 3), (4, 4, 4, 256, 64, 3),
            (2, 16, 16, 64, 192, 1), (1, 13, 9, 8, 3, 3), (8, 32, 32, 128, 128, 3), (2, 8, 8, 4, 32, 3),
            # the layer shapes of the 237 M-parameter UNet at a 64x64 latent, batch 2 (split-K tiles, 4-16 Cout tiles)
            (2, 16, 16, 1024, 1024, 3), (2, 16, 16, 2048, 1024, 3), (2, 32, 32, 1536, 512, 3), (2, 64, 64, 640, 128, 3),
            (2, 16, 16, 2048, 1024, 1), (2, 64, 64, 128, 128, 3), (2, 32, 32, 512, 512, 3),
            # more 1x1 weight gradients (TN GEMM over the pixels): ragged channel tiles, K splits, a short K
            (4, 16, 16, 160, 132, 1), (32, 16, 16, 1024, 512, 1), (3, 8, 8, 96, 72, 1),
            # 1x1 weight gradients on the bf16x3 pipe kernel behind the transposing split pass (K >= 4096 pixels, Cin % 32 == 0; the
            # default takes it from ~100 FLOP per split byte: the last shape; all of them in the BBDM_WGRAD1X1_BF3=2 child below): Cout
            # below / not a multiple of the 128-column tile, a pixel count that is not a multiple of 256 (zero rows enter the contraction)
            (4, 32, 32, 64, 72, 1), (1, 72, 60, 96, 256, 1), (2, 64, 64, 256, 128, 1), (5, 32, 32, 1024, 1536, 1),
            # 3x3 layers with a thin side on the vector-ALU kernel (conv_wgrad_thin_f32): the stem / head of the UNet, ragged tiles,
            # more than one 128-channel tile on the wide side, 1..4 output channels
            (2, 64, 64, 4, 128, 3), (2, 64, 64, 128, 3, 3), (2, 20, 12, 4, 132, 3), (1, 9, 9, 260, 4, 3), (3, 8, 8, 16, 1, 3)]


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks", CONV_BWD)
def test_conv_backward(dev, N, H, W, Cin, Cout, ks):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N + H + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * 0.05).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    dy = torch.randn(N, Cout, H, W, generator=g)
    F.conv2d(x, w, b, padding=ks // 2).backward(dy)
    cpad = (Cout + 3) // 4 * 4
    dyg = torch.zeros(N, H, W, cpad)
    dyg[..., :Cout] = _nhwc(dy)
    dyg = dyg.to(dev)
    xg = _nhwc(x.detach()).to(dev)
    # data gradient
    pwd = ops.pack_conv_weight_dgrad(w.detach().to(dev), cout_in=cpad)
    dx = ops.conv2d_nhwc(dyg, pwd, None, Cin, ks)
    assert rel_err(_nchw(dx.cpu()), x.grad) < TOL
    # weight / bias gradient
    dw, db2 = ops.conv_wgrad(xg, dyg, Cin, Cout, ks, with_bias=True)
    db = ops.colsum(dyg, Cout)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < TOL
    assert rel_err(db.cpu(), b.grad) < TOL and rel_err(db2.cpu(), b.grad) < TOL


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks", [c for c in CONV_BWD if c[5] == 1])
def test_conv1x1_wgrad_planes_path(dev, N, H, W, Cin, Cout, ks):
    """Option "wgrad1x1_bf3" = 2 (read at every call): every 1x1 shape the plane layout accepts takes the transposing split pass +
    bbdm_gemm_bf3p_tn_f32 instead of gemm_tn_f32 -- the small ragged cases of CONV_BWD included; 0: always gemm_tn_f32."""
    from bbdm_amd import _lib
    for mode in (2, 0):
        with _lib.option("wgrad1x1_bf3", mode):
            test_conv_backward(dev, N, H, W, Cin, Cout, ks)


def test_conv_wgrad_checks_its_workspace(dev):
    """bbdm_conv_wgrad_f32 receives the size of its workspace (ABI 21).  A workspace sized for the TN GEMM but too small for the
    bf16-plane path (a caller that sized it under another "wgrad1x1_bf3" setting) falls back to the TN GEMM with the same result;
    any other shortfall is refused before anything is launched."""
    import kernel_ops as ops
    from bbdm_amd import _lib
    lib = _lib.load()
    N, H, W, Cin, Cout = 5, 32, 32, 1024, 1536                 # takes the plane path by default
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(dev)
    with _lib.option("wgrad1x1_bf3", 0):
        small = lib.bbdm_conv_wgrad_workspace_floats(N, H, W, Cin, Cout, 1)
        want, wantb = ops.conv_wgrad(x, dy, Cin, Cout, 1, with_bias=True)
    full = lib.bbdm_conv_wgrad_workspace_floats(N, H, W, Cin, Cout, 1)
    assert small < full, (small, full)
    got, gotb = ops.conv_wgrad(x, dy, Cin, Cout, 1, with_bias=True, ws_floats=small)       # planes refused -> TN GEMM
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(gotb, wantb)
    with pytest.raises(_lib.BBDMHipError, match="workspace"):
        ops.conv_wgrad(x, dy, Cin, Cout, 1, ws_floats=64)
    with pytest.raises(_lib.BBDMHipError, match="workspace"):
        ops.conv_wgrad(x[:1, :8, :8, :64].contiguous(), dy[:1, :8, :8, :32].contiguous(), 64, 32, 3, ws_floats=16)


# (m, N, H, W, Cin, Cout): the Winograd-domain weight gradient (csrc/winograd_wgrad.hip) incl. ragged m = 6 tiles, K splits,
# both GEMM tile heights (Cin % 256 == 0 or not) and the LBBDM-f4 training layer shapes at batch 2
WINO_WGRAD = [(2, 2, 8, 8, 16, 24), (4, 1, 8, 12, 32, 64), (6, 2, 12, 12, 64, 128), (6, 1, 7, 10, 16, 8), (6, 3, 16, 20, 48, 72),
              (4, 2, 16, 16, 256, 132), (6, 2, 64, 64, 128, 128), (6, 2, 64, 64, 640, 128), (4, 2, 32, 32, 512, 512),
              (4, 2, 32, 32, 1536, 512), (4, 2, 16, 16, 1024, 1024), (4, 2, 16, 16, 2048, 1024), (6, 8, 64, 64, 256, 128)]


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", WINO_WGRAD)
def test_winograd_wgrad(dev, m, N, H, W, Cin, Cout):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(m + N + H + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, b, padding=1).backward(dy.double())              # fp64 reference gradient
    pitch = Cout + 8                                                         # dy as a channel slice of a wider buffer
    dyg = torch.randn(N, H, W, pitch, generator=g)
    dyg[..., :Cout] = _nhwc(dy)
    dw, db = ops.conv3x3_winograd_wgrad(_nhwc(x).to(dev), dyg.to(dev), Cout, m, with_bias=True)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    e = rel_err(dw.cpu(), w.grad.float())
    print(f"winograd wgrad m={m} N{N} {H}x{W} {Cin}->{Cout}: rel err {e:.2e}")
    # against the fp64 gradient; fp32 Winograd rounding grows with the tile (tests/test_winograd_math_cpu.py) and with the
    # number of pixels summed: measured 3e-7 (m = 2) ... 2.8e-5 (m = 6, 8 x 64 x 64 pixels) -- gradient bar: 1e-3
    assert e < 1e-4
    assert rel_err(db.cpu(), b.grad.float()) < TOL


# the same gradient with both GEMM operands as transposed bf16 planes on the bf16x3 kernel (Cin % 32 == 0): ragged m = 6 tiles, padded
# tile counts (zeros must enter the contraction), Cout below / not a multiple of the 128-column tile, split-K, the LBBDM-f4 layer shapes
WINO_WGRAD_BF3P = [(2, 2, 8, 8, 32, 24), (4, 1, 8, 12, 32, 64), (6, 2, 12, 12, 64, 128), (6, 1, 7, 10, 32, 8), (6, 3, 16, 20, 96, 72),
                   (4, 2, 16, 16, 256, 132), (6, 2, 64, 64, 128, 128), (6, 2, 64, 64, 640, 128), (4, 2, 32, 32, 512, 512),
                   (4, 2, 32, 32, 1536, 512), (4, 2, 16, 16, 1024, 1024), (6, 8, 64, 64, 256, 128),
                   # m = 8 (round 5, UNetModel.winograd_train8): ragged / padded tiles, Cout not a multiple of 128, the C4 layer shapes
                   (8, 2, 16, 24, 32, 24), (8, 1, 13, 18, 64, 132), (8, 2, 64, 64, 128, 128), (8, 2, 64, 64, 640, 128),
                   (8, 2, 32, 32, 512, 512), (8, 2, 32, 32, 1536, 512), (8, 8, 64, 64, 256, 128)]


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", WINO_WGRAD_BF3P)
def test_winograd_wgrad_bf3p(dev, m, N, H, W, Cin, Cout):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(m + N + H + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, b, padding=1).backward(dy.double())              # fp64 reference gradient
    pitch = Cout + 8
    dyg = torch.randn(N, H, W, pitch, generator=g)
    dyg[..., :Cout] = _nhwc(dy)
    dw, db = ops.conv3x3_winograd_wgrad_bf3p(_nhwc(x).to(dev), dyg.to(dev), Cout, m, with_bias=True)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    e = rel_err(dw.cpu(), w.grad.float())
    print(f"winograd wgrad (bf16x3 TN GEMM) m={m} N{N} {H}x{W} {Cin}->{Cout}: rel err {e:.2e}")
    assert e < (5e-4 if m == 8 else 1e-4)       # (m = 8: ~6x m = 6's rounding, tests/test_winograd_math_cpu.py; gradient bar 1e-3)
    assert rel_err(db.cpu(), b.grad.float()) < TOL


@pytest.mark.parametrize("batch,K,M,N", [(3, 40, 64, 36), (2, 1000, 256, 128), (36, 512, 132, 260), (1, 5000, 128, 128)])
def test_gemm_tn_batched(dev, batch, K, M, N):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(batch + K + M + N)
    a = torch.randn(batch, K, M, generator=g)
    b = torch.randn(batch, K, N, generator=g)
    want = torch.einsum("bkm,bkn->bmn", a.double(), b.double()).float()
    got = ops.gemm_tn_batched(a.to(dev), b.to(dev))
    assert rel_err(got.cpu(), want) < 1e-5


GN_BWD = [(2, 16, 16, 128), (1, 8, 8, 640), (2, 4, 4, 1536), (2, 8, 8, 96), (3, 16, 16, 32), (2, 32, 32, 256),
          (2, 16, 16, 2048), (2, 32, 32, 1536), (2, 64, 64, 640), (2, 16, 16, 1024)]      # full-size UNet layers


@pytest.mark.parametrize("N,H,W,C", GN_BWD)
@pytest.mark.parametrize("mode", ["plain", "film_silu", "silu_pool", "silu_up", "silu_add_acc"])
def test_groupnorm_backward(dev, N, H, W, C, mode):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(C + H + len(mode))
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).requires_grad_()
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_()
    film = (0.3 * torch.randn(N, 2 * C, generator=g)).requires_grad_()
    y = F.group_norm(x, 32, gamma, beta, 1e-5)
    resample, silu, use_film = 0, mode != "plain", mode == "film_silu"
    if use_film:
        y = y * (1 + film[:, :C, None, None]) + film[:, C:, None, None]
    if silu:
        y = F.silu(y)
    if mode == "silu_pool":
        y, resample = F.avg_pool2d(y, 2, 2), 1
    elif mode == "silu_up":
        y, resample = F.interpolate(y, scale_factor=2, mode="nearest"), 2
    da = torch.randn(y.shape, generator=g)
    dadd = torch.randn(y.shape, generator=g) if mode in ("silu_add_acc", "silu_pool", "silu_up") else None
    extra = x
    if mode == "silu_pool":
        extra = F.avg_pool2d(x, 2, 2)
    elif mode == "silu_up":
        extra = F.interpolate(x, scale_factor=2, mode="nearest")
    loss = (y * da).sum() + ((extra * dadd).sum() if dadd is not None else 0.0)
    loss.backward()
    xg = _nhwc(x.detach()).to(dev)
    stats = ops.groupnorm_stats(xg)
    init = torch.randn(N, H, W, C, generator=g) if mode == "silu_add_acc" else None
    dx0 = init.clone().to(dev) if init is not None else None
    dx, dg, dbt, dfilm = ops.groupnorm_bwd(xg, stats, gamma.detach().to(dev), beta.detach().to(dev), _nhwc(da).to(dev),
                                           film=film.detach().to(dev) if use_film else None,
                                           dadd=_nhwc(dadd).to(dev) if dadd is not None else None, silu=silu,
                                           resample=resample, dx=dx0, accumulate=init is not None)
    torch.cuda.synchronize()
    want = x.grad + (_nchw(init) if init is not None else 0.0)
    assert rel_err(_nchw(dx.cpu()), want) < TOL
    assert rel_err(dg.cpu(), gamma.grad) < TOL and rel_err(dbt.cpu(), beta.grad) < TOL
    if use_film:
        assert rel_err(dfilm.cpu(), film.grad) < TOL


def test_resample_only_backward(dev):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 8, 12, generator=g, requires_grad=True)
    for rs, fn in ((1, lambda t: F.avg_pool2d(t, 2, 2)), (2, lambda t: F.interpolate(t, scale_factor=2, mode="nearest"))):
        x.grad = None
        y = fn(x)
        d = torch.randn(y.shape, generator=g)
        y.backward(d)
        dx, _, _, _ = ops.groupnorm_bwd(_nhwc(x.detach()).to(dev), None, None, None, None, dadd=_nhwc(d).to(dev),
                                        resample=rs)
        torch.cuda.synchronize()
        assert rel_err(_nchw(dx.cpu()), x.grad) < 1e-6


ATTN_BWD = [(2, 16, 4, 64), (1, 256, 2, 64), (2, 100, 3, 32), (3, 16, 2, 16), (1, 300, 1, 64), (2, 256, 16, 64)]


@pytest.mark.parametrize("N,T,heads,ch", ATTN_BWD)
@pytest.mark.parametrize("new_order", [False, True])
def test_attention_backward(dev, N, T, heads, ch, new_order):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(T + heads + ch)
    C = heads * ch
    qkv = (torch.randn(N, 3 * C, T, generator=g) * 1.2).requires_grad_()
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(N * heads, ch, T) for z in (q, k, v))
    else:
        q, k, v = qkv.reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v).reshape(N, C, T)
    do = torch.randn(N, C, T, generator=g)
    ref.backward(do)
    qg = qkv.detach().permute(0, 2, 1).contiguous().to(dev)
    out, lse = ops.attention(qg, heads, new_order, return_lse=True)
    dqkv = ops.attention_bwd(qg, out, do.permute(0, 2, 1).contiguous().to(dev), lse, heads, new_order)
    torch.cuda.synchronize()
    assert rel_err(out.cpu().permute(0, 2, 1), ref.detach()) < TOL
    assert rel_err(dqkv.cpu().permute(0, 2, 1), qkv.grad) < TOL


@pytest.mark.parametrize("N,In,Out,act", [(4, 128, 512, False), (16, 512, 1000, True), (33, 96, 70, True),
                                           (64, 512, 2048, True), (2, 512, 25088, True)])
def test_linear_backward(dev, N, In, Out, act):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N + In)
    x = torch.randn(N, In, generator=g, requires_grad=True)
    w = (torch.randn(Out, In, generator=g) * 0.05).requires_grad_()
    b = torch.randn(Out, generator=g).requires_grad_()
    dy = torch.randn(N, Out, generator=g)
    F.linear(F.silu(x) if act else x, w, b).backward(dy)
    dx, dw, db = ops.linear_bwd(dy.to(dev), x.detach().to(dev), w.detach().to(dev), act_in=act)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu(), x.grad) < TOL and rel_err(dw.cpu(), w.grad) < TOL and rel_err(db.cpu(), b.grad) < TOL


# ---- SpatialTransformer pieces (attention.py:153-219): cross-attention, LayerNorm, GEGLU -------------------------------------
@pytest.mark.parametrize("N,Tq,Tk,heads,ch", [(2, 64, 64, 2, 32), (1, 200, 77, 4, 64), (2, 130, 260, 1, 16), (1, 1024, 256, 8, 64)])
def test_cross_attention_backward(dev, N, Tq, Tk, heads, ch):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N + Tq + Tk + heads)
    C = heads * ch
    q = torch.randn(N, Tq, C, generator=g).requires_grad_()
    k = torch.randn(N, Tk, C, generator=g).requires_grad_()
    v = torch.randn(N, Tk, C, generator=g).requires_grad_()
    dout = torch.randn(N, Tq, C, generator=g)
    split = lambda t: t.view(N, -1, heads, ch).permute(0, 2, 1, 3)            # 'b n (h d) -> b h n d'
    sim = torch.einsum("bhid,bhjd->bhij", split(q), split(k)) * ch ** -0.5   # attention.py:178-190
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), split(v)).permute(0, 2, 1, 3).reshape(N, Tq, C)
    ref.backward(dout)
    out, lse = ops.cross_attention(q.detach().to(dev), k.detach().to(dev), v.detach().to(dev), heads, return_lse=True)
    dq, dk, dv = ops.cross_attention_bwd(q.detach().to(dev), k.detach().to(dev), v.detach().to(dev), out, lse, dout.to(dev), heads)
    assert rel_err(out.cpu(), ref.detach()) < TOL
    assert rel_err(dq.cpu(), q.grad) < TOL and rel_err(dk.cpu(), k.grad) < TOL and rel_err(dv.cpu(), v.grad) < TOL


@pytest.mark.parametrize("rows,C,with_add", [(37, 64, False), (512, 320, True), (4096, 1024, True)])
def test_layernorm_backward(dev, rows, C, with_add):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 1.7 + 0.4).requires_grad_()
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_()
    dy = torch.randn(rows, C, generator=g)
    dadd = torch.randn(rows, C, generator=g) if with_add else None
    F.layer_norm(x, (C,), gamma, beta, 1e-5).backward(dy)
    dx, dg, db = ops.layernorm_bwd(x.detach().to(dev), gamma.detach().to(dev), dy.to(dev), 1e-5,
                                   None if dadd is None else dadd.to(dev))
    want = x.grad + (dadd if dadd is not None else 0.0)
    assert rel_err(dx.cpu(), want) < TOL
    assert rel_err(dg.cpu(), gamma.grad) < TOL and rel_err(db.cpu(), beta.grad) < TOL


@pytest.mark.parametrize("rows,inner", [(33, 32), (1024, 1280)])
def test_geglu_backward(dev, rows, inner):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(rows + inner)
    a = (torch.randn(rows, 2 * inner, generator=g) * 1.5).requires_grad_()
    dy = torch.randn(rows, inner, generator=g)
    (a[:, :inner] * F.gelu(a[:, inner:])).backward(dy)
    da = ops.geglu_bwd(a.detach().to(dev), dy.to(dev))
    assert rel_err(da.cpu(), a.grad) < TOL
