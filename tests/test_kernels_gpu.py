"""Kernel-level parity (GPU): every C-ABI entry point against the same op in plain PyTorch fp32 on the CPU
(F.conv2d, F.group_norm, softmax attention, F.linear, the scheduler formulas of the oracle).

Tolerances: fp32 arithmetic with a different summation order -> 2e-5 relative to the output's max magnitude
(the north-star bar is 1e-3 per sampling step for the whole UNet)."""
import math

import pytest
import torch
import torch.nn.functional as F

import bbdm_oracle as O
from fixtures import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # N, H, W, Cin, Cout, ks, residual
    (2, 16, 16, 32, 64, 3, False),
    (1, 64, 64, 128, 128, 3, True),      # 256-pixel tiles, TW=32
    (3, 8, 8, 64, 128, 3, True),         # several images per block
    (5, 4, 4, 96, 160, 3, False),        # tiny images, ragged N, Cout not a multiple of 128
    (2, 12, 20, 48, 40, 3, True),        # non power-of-two H, W; Cin not a multiple of 16
    (1, 33, 7, 8, 3, 3, False),          # odd sizes, Cout = 3
    (2, 16, 16, 256, 128, 1, True),      # 1x1 (skip connection / qkv / proj)
    (4, 2, 2, 64, 192, 1, False),
    (1, 32, 32, 6, 128, 3, False),       # first conv: Cin = 6 (padded to 8 by the layout kernel)
    (16, 32, 32, 128, 128, 3, True),     # >= 512 blocks -> 256x128 tile path
    (3, 2, 2, 64, 64, 3, True),          # 2x2 images: images-per-block capped by the patch slots
    (70, 1, 1, 32, 32, 3, False),        # 1x1 images
    (40, 4, 4, 128, 1024, 3, False),     # 4x4 latents (LBBDM-f16 bottom level)
    (32, 4, 4, 1024, 1024, 3, True),     # split-K: 32 output tiles, 64 chunks
    (4, 16, 16, 2048, 512, 1, True),     # split-K on a 1x1
    (2, 8, 8, 640, 96, 3, False),        # split-K with ragged Cout
    (1, 64, 64, 32, 3, 3, False),        # the UNet head: few output channels -> one-thread-per-pixel kernel
    (2, 48, 50, 20, 8, 3, False),        # same, ragged tiles, Cin not a multiple of 16, Cout = 8
    (1, 72, 60, 6, 128, 3, False),       # the UNet stem from 4096 pixels: K = 72 in one stage (conv3x3_stem_kernel), ragged 16x16 tiles
    (3, 64, 64, 8, 128, 3, False),       # ... several images, whole tiles
    (2, 72, 40, 3, 128, 3, False),       # ... the latent UNets' / the VQGAN encoder's 3 -> 128 (4 padded channels, K = 36)
    (5, 40, 24, 144, 4, 3, False),       # small-image head (8x8 tiles, chunks split over the waves): 9 chunks = a ragged last round
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,res", CONV_CASES)
def test_conv2d(dev, N, H, W, Cin, Cout, ks, res):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    r = torch.randn(N, Cout, H, W, generator=g) if res else None
    ref = F.conv2d(x, w, b, padding=ks // 2)
    if res:
        ref = ref + r
    cpad = (Cin + 3) // 4 * 4
    xg = ops.nchw_to_nhwc(x.to(dev), cpad=cpad)
    pw = ops.pack_conv_weight(w.to(dev), cin_pad=cpad)
    rg = _nhwc(r).to(dev) if res else None
    out = ops.conv2d_nhwc(xg, pw, b.to(dev), Cout, ks, residual=rg)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref) < TOL
    # NCHW epilogue + in-place residual (out aliases residual)
    out2 = ops.conv2d_nhwc(xg, pw, b.to(dev), Cout, ks, residual=None, out_nchw=True)
    ref2 = F.conv2d(x, w, b, padding=ks // 2)
    assert rel_err(out2.cpu(), ref2) < TOL
    if res:
        buf = rg.clone()
        ops.conv2d_nhwc(xg, pw, b.to(dev), Cout, ks, residual=buf, out=buf)
        assert rel_err(_nchw(buf.cpu()), ref) < TOL


WINO_CASES = [
    # m, N, H, W, Cin, Cout, residual (0 none, 1 per pixel, 2 per image)
    (2, 2, 16, 16, 64, 128, 0),
    (2, 1, 32, 32, 256, 256, 1),
    (2, 3, 8, 12, 48, 72, 1),            # ragged tile count (padded to whole GEMM tiles), Cin not a multiple of 16
    (2, 2, 64, 64, 320, 384, 2),
    (2, 1, 2, 2, 16, 8, 0),              # a single tile
    (2, 5, 6, 4, 132, 260, 1),
    (4, 2, 16, 16, 64, 128, 0),
    (4, 1, 32, 32, 256, 256, 1),
    (4, 3, 8, 12, 48, 72, 1),
    (4, 2, 64, 64, 320, 384, 2),
    (4, 1, 4, 4, 16, 8, 0),
    (4, 5, 12, 4, 132, 260, 1),
    (4, 1, 32, 32, 1024, 512, 1),        # long reduction: rounding error grows with sqrt(Cin)
]
# rel_err is max|diff| / max|ref|.  F(2x2,3x3) transforms use 0, +-1, +-1/2 only; F(4x4,3x3) uses up to 8 (+ 1/24 in
# the weights) and is ~10x noisier (csrc/winograd.hip header) -- both orders of magnitude inside the 1e-3 step bar.
# m = 8 (ten points): measured 1 - 2.5e-4 per layer on the bf16x3 planes, 0.6 - 1.5e-4 on the fp16-pair planes (profiles/r06_h2_probe*.txt);
# 5e-4 = half the whole STEP's bar, so that a 2x regression of the transform or the GEMM shows here (round-5 verdict)
WINO_TOL = {2: 2e-5, 4: 1e-4, 6: 2e-4, 7: 2e-4, 8: 5e-4}


@pytest.mark.parametrize("m,N,H,W,Cin,Cout,res", WINO_CASES)
def test_conv3x3_winograd(dev, m, N, H, W, Cin, Cout, res):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    rg = None
    if res == 1:
        r = torch.randn(N, Cout, H, W, generator=g)
        ref = ref + r
        rg = _nhwc(r).to(dev)
    elif res == 2:
        r = torch.randn(N, Cout, generator=g)
        ref = ref + r[:, :, None, None]
        rg = r.to(dev)
    xg = ops.nchw_to_nhwc(x.to(dev), cpad=Cin)
    pw = ops.pack_winograd_weight(w.to(dev), in_pad=Cin, m=m)
    out = ops.conv3x3_winograd(xg, pw, b.to(dev), Cout, residual=rg, res_per_image=(res == 2), m=m)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref.float()) < WINO_TOL[m]
    # and it agrees with the direct implicit-GEMM kernel on the same input
    direct = ops.conv2d_nhwc(xg, ops.pack_conv_weight(w.to(dev), cin_pad=Cin), b.to(dev), Cout, 3)
    if res == 1:
        direct = direct + rg
    elif res == 2:
        direct = direct + rg[:, None, None, :]
    assert rel_err(out.cpu(), direct.cpu()) < WINO_TOL[m]
    if res == 1:                      # in place: out aliases the residual
        buf = rg.clone()
        ops.conv3x3_winograd(xg, pw, b.to(dev), Cout, residual=buf, out=buf, m=m)
        assert rel_err(_nchw(buf.cpu()), ref.float()) < WINO_TOL[m]


@pytest.mark.parametrize("m", [2, 4])
def test_conv3x3_winograd_dgrad_and_slices(dev, m):
    """Data gradient through the dgrad packing; input / output as channel slices of wider buffers."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(11)
    N, H, W, Cin, Cout = 2, 16, 16, 96, 160
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    dy = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1).float()          # dX of conv2d(x, w, padding=1)
    wide_in = torch.zeros(N, H, W, 200, device=dev)
    wide_in[..., 40:200] = _nhwc(dy).to(dev)
    wide_out = torch.zeros(N, H, W, 128, device=dev)
    pw = ops.pack_winograd_weight(w.to(dev), in_pad=Cout, dgrad=True, m=m)
    lib = _lib.load()
    ws = torch.empty(lib.bbdm_winograd_workspace_floats(m, N, H, W, Cout, Cin), device=dev)
    _lib.call("bbdm_conv3x3_winograd_f32", m, wide_in.data_ptr() + 4 * 40, 200, pw.data_ptr(), None, None, 0,
              wide_out.data_ptr() + 4 * 16, 128, 0, ws.data_ptr(), N, H, W, Cout, Cin,
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = wide_out.cpu()
    assert rel_err(_nchw(got[..., 16:112]), ref) < WINO_TOL[m]
    assert float(got[..., :16].abs().max()) == 0 and float(got[..., 112:].abs().max()) == 0


@pytest.mark.parametrize("m,up,silu", [(2, 0, 1), (2, 1, 0), (2, 1, 1), (2, 0, 0), (4, 0, 1), (4, 1, 1), (4, 1, 0)])
def test_winograd_stages_fused_producer(dev, m, up, silu):
    """The three stages called separately, with the GroupNorm/FiLM/SiLU producer and the nearest x2 upsampling folded
    into the input transform (zero padding applies to the activated, upsampled tensor)."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(21 + 2 * up + silu)
    N, H, W, Cin, Cout = 3, 16, 24, 64, 96                    # conv resolution; the source is H/2 x W/2 when up
    hs, ws_ = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(N, Cin, hs, ws_, generator=g)
    sc = torch.randn(N, Cin, generator=g)
    bi = torch.randn(N, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    a = x.double() * sc.double()[:, :, None, None] + bi.double()[:, :, None, None]
    if silu:
        a = F.silu(a)
    if up:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    ref = F.conv2d(a, w.double(), b.double(), padding=1).float()
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    xg, scg, big, bg = _nhwc(x).to(dev), sc.to(dev), bi.to(dev), b.to(dev)
    pw = ops.pack_winograd_weight(w.to(dev), m=m)
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    assert tiles % 256 == 0 and tiles >= N * (H // m) * (W // m)
    P = (m + 2) ** 2
    V = torch.empty(P * tiles * Cin, device=dev)
    M = torch.empty(P * tiles * Cout, device=dev)
    out = torch.empty(N, H, W, Cout, device=dev)
    _lib.call("bbdm_winograd_input_f32", m, xg.data_ptr(), Cin, V.data_ptr(), scg.data_ptr(), big.data_ptr(), Cin, silu, up,
              N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_f32", m, V.data_ptr(), pw.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
    _lib.call("bbdm_winograd_output_f32", m, M.data_ptr(), bg.data_ptr(), None, 0, out.data_ptr(), Cout, 0, N, H, W, Cout, st)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref) < WINO_TOL[m]


# m = 6: first run on hardware in round 2 (tools/ab_winograd6.sh: 11 passed; adopted for the layers where it wins)
@pytest.mark.parametrize("N,H,W,Cin,Cout,res", [
    (2, 12, 12, 64, 128, 0), (1, 24, 36, 256, 256, 1), (2, 64, 64, 320, 384, 2),      # 64 is not a multiple of 6
    (3, 16, 20, 48, 72, 1), (1, 6, 6, 16, 8, 0), (1, 32, 32, 1024, 512, 1), (5, 7, 9, 132, 260, 1)])
def test_conv3x3_winograd_m6(dev, N, H, W, Cin, Cout, res):
    test_conv3x3_winograd(dev, 6, N, H, W, Cin, Cout, res)


@pytest.mark.parametrize("up,silu", [(0, 1), (1, 1), (1, 0)])
def test_winograd_stages_fused_producer_m6(dev, up, silu):
    test_winograd_stages_fused_producer(dev, 6, up, silu)


def test_conv3x3_winograd_dgrad_and_slices_m6(dev):
    test_conv3x3_winograd_dgrad_and_slices(dev, 6)


@pytest.mark.parametrize("batch,T,Cin,Cout", [(2, 256, 48, 72), (1, 512, 256, 128), (3, 256, 1024, 260)])
def test_gemm_bf3_accuracy(dev, batch, T, Cin, Cout):
    """csrc/gemm_bf3.hip: the fp32 GEMM on the BF16 matrix core (exact three-way operand split, six product terms) against
    an fp64 GEMM, next to plain fp32 arithmetic on the same operands: it must be as accurate as fp32 (not bf16: a single
    bf16 product would be off by ~4e-3)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(Cin + Cout)
    V = torch.randn(batch, T, Cin, generator=g) * torch.logspace(-3, 3, Cin)          # 6 decades of column scales
    Wt = torch.randn(batch, Cout, Cin, generator=g) * 0.1 / torch.logspace(-3, 3, Cin)
    cout_pad = (Cout + 127) // 128 * 128
    pk = torch.zeros(batch, Cin // 16, cout_pad, 16)
    for c in range(Cin // 16):
        pk[:, c, :Cout, :] = Wt[:, :, c * 16:(c + 1) * 16]
    M = ops.gemm_bf3(V.to(dev), pk.contiguous().to(dev), batch, Cin, Cout).cpu()
    torch.cuda.synchronize()
    ref = torch.einsum("btk,bok->bto", V.double(), Wt.double())
    f32 = torch.einsum("btk,bok->bto", V, Wt)
    e_bf3, e_f32 = rel_err(M, ref), rel_err(f32, ref)
    print(f"gemm_bf3 [{batch} x {T} x {Cin} x {Cout}]: rel err vs fp64 {e_bf3:.2e} (plain fp32: {e_f32:.2e})")
    assert e_bf3 < 3e-6 and e_bf3 < 8 * e_f32 + 1e-7


@pytest.mark.parametrize("kernel", [4, 5, 7])
@pytest.mark.parametrize("batch,T,Cin,Cout", [(2, 512, 16, 256), (1, 256, 32, 512), (8, 256, 80, 260), (3, 768, 1024, 256),
                                              (2, 768, 48, 128), (8, 1280, 128, 72)])
def test_gemm_bf3p_kernel_variants(dev, kernel, batch, T, Cin, Cout):
    """Every tile shape of csrc/gemm_bf3p.hip's kernel (option "bf3p_kernel" of the header's options section: 4 / 5 / 7 = 256 x 256,
    256 x 128, 128 x 128 workgroup tiles; the default picks among them by the number of workgroups they give, which on test-size
    problems is always the smallest) is bit-equal to csrc/gemm_bf3.hip; Cin = 16 / 32 are the one- and two-chunk edge cases of the prologues, Cout = 260 falls back
    from the 256-column tiles, T = 768 / 1280 leave the 512-row tiles a ragged last row tile."""
    from bbdm_amd import _lib
    with _lib.option("bf3p_kernel", kernel):
        test_gemm_bf3p_matches_bf3_bitwise(dev, batch, T, Cin, Cout, 0)
        test_gemm_bf3p_matches_bf3_bitwise(dev, batch, T, Cin, Cout, 2)


@pytest.mark.parametrize("batch,T,Cin,Cout,extra", [(2, 256, 48, 72, 0), (1, 512, 256, 128, 1), (3, 256, 1024, 260, 2),
                                                    (8, 300, 64, 132, 0), (16, 768, 32, 256, 0),
                                                    (36, 512, 32, 256, 0), (12, 768, 32, 132, 2), (9, 300, 32, 72, 1), (10, 256, 32, 72, 0),
                                                    (11, 256, 32, 40, 0), (100, 256, 16, 128, 0)])
def test_gemm_bf3p_matches_bf3_bitwise(dev, batch, T, Cin, Cout, extra):
    """csrc/gemm_bf3p.hip (operands pre-split by their producers, LDS-DMA staging) must reproduce csrc/gemm_bf3.hip BIT FOR BIT --
    same exact split, same six terms in the same order -- and hence its fp32-class accuracy against an fp64 GEMM.  extra: 1 = bias,
    2 = bias + residual (in place).  batch 8 / 16 take the each-XCD-owns-whole-entries launch; T = 300 the zero-padded rows; batch 36 /
    12 / 9 / 10 / 100 end in a group of 4 / 4 / 1 / 2 / 4 entries that the XCDs SHARE (every (8 / rem)-th tile each, one or several tiles
    per entry), batch 11 in one of 3 that keeps whole entries."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(Cin + Cout + T)
    V = torch.randn(batch, T, Cin, generator=g) * torch.logspace(-3, 3, Cin)
    Wt = torch.randn(batch, Cout, Cin, generator=g) * 0.1 / torch.logspace(-3, 3, Cin)
    cout_pad = (Cout + 127) // 128 * 128
    pk = torch.zeros(batch, Cin // 16, cout_pad, 16)
    for c in range(Cin // 16):
        pk[:, c, :Cout, :] = Wt[:, :, c * 16:(c + 1) * 16]
    pk = pk.contiguous().to(dev)
    bias = torch.randn(Cout, generator=g) if extra else None
    res = torch.randn(batch, T, Cout, generator=g) if extra == 2 else None
    M = ops.gemm_bf3p(V.to(dev), pk, batch, Cin, Cout, None if bias is None else bias.to(dev),
                      None if res is None else res.to(dev)).cpu()
    torch.cuda.synchronize()
    ref = torch.einsum("btk,bok->bto", V.double(), Wt.double())
    if bias is not None:
        ref = ref + bias.double()
    if res is not None:
        ref = ref + res.double()
    e = rel_err(M, ref)
    print(f"gemm_bf3p [{batch} x {T} x {Cin} x {Cout}]: rel err vs fp64 {e:.2e}")
    assert e < 3e-6
    if T % 256 == 0 and not extra:
        M0 = ops.gemm_bf3(V.to(dev), pk, batch, Cin, Cout).cpu()
        torch.cuda.synchronize()
        assert torch.equal(M, M0), (M - M0).abs().max()


def _pack_b(Wt, Cin, Cout):
    """[batch, Cout, Cin] -> the fp32 packed layout [batch][Cin / 16][CoutPad128][16] of the plane GEMMs' B operand."""
    batch = Wt.shape[0]
    cout_pad = (Cout + 127) // 128 * 128
    pk = torch.zeros(batch, Cin // 16, cout_pad, 16)
    for c in range(Cin // 16):
        pk[:, c, :Cout, :] = Wt[:, :, c * 16:(c + 1) * 16]
    return pk.contiguous()


@pytest.mark.parametrize("K", [128, 512, 2048])
def test_gemm_h2p_accuracy(dev, K):
    """The fp16-pair planes (csrc/h2_split.h: two fp16 planes per operand under a power-of-two scale, three MFMA terms) against an fp64
    GEMM, next to the six-term bf16x3 planes on the same operands: HALF the roundings of the fp32 accumulator, so the error is LOWER
    from K = 128 up (measured MI355X, rms: K = 128 1.45e-7 / 1.64e-7, 512 2.6e-7 / 3.4e-7, 2048 5.1e-7 / 7.0e-7;
    profiles/r06_gemm_error_{h2,bf3}.txt) -- the bar here is 'no worse than bf16x3 and fp32-class'."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(K)
    batch, T, Cout = 8, 512, 128
    V = torch.randn(batch, T, K, generator=g)
    Wt = torch.randn(batch, Cout, K, generator=g) * 0.05
    pk = _pack_b(Wt, K, Cout).to(dev)
    ref = torch.einsum("btk,bok->bto", V.double(), Wt.double())
    rms = lambda m: float(((m.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    Mh = ops.gemm_h2p(V.to(dev), pk, batch, K, Cout).cpu()
    Mb = ops.gemm_bf3p(V.to(dev), pk, batch, K, Cout).cpu()
    eh, eb = rms(Mh), rms(Mb)
    print(f"gemm K={K}: rms vs fp64 h2 {eh:.2e} bf16x3 {eb:.2e}; max-norm h2 {rel_err(Mh, ref):.2e} bf16x3 {rel_err(Mb, ref):.2e}")
    assert eh < 1.0e-6 and eh <= 1.05 * eb and rel_err(Mh, ref) < 3e-6


@pytest.mark.parametrize("batch,T,Cin,Cout,extra,decades", [(2, 256, 48, 72, 0, 3), (1, 512, 256, 128, 1, 3), (3, 256, 1024, 260, 2, 3),
                                                            (8, 300, 64, 132, 0, 3), (36, 512, 32, 256, 0, 0), (100, 256, 16, 128, 0, 3),
                                                            (9, 300, 32, 72, 1, 6)])
def test_gemm_h2p_shapes_and_dynamic_range(dev, batch, T, Cin, Cout, extra, decades):
    """The shapes of test_gemm_bf3p_matches_bf3_bitwise on the fp16-pair planes (bias, residual in place, ragged rows, shared XCD
    groups), with the operands' magnitudes spread over ``decades`` decades ACROSS the contraction index.  Up to ~4 decades below the
    operand's maximum both planes are normal fp16 numbers (22 - 23 significant bits); below that the second plane turns subnormal and the
    element keeps fewer bits -- 6 decades cost an order of magnitude (the product path only feeds this GEMM tensors whose range a
    GroupNorm bounds: DESIGN.md §2)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(Cin + Cout + T)
    spread = torch.logspace(-decades / 2, decades / 2, Cin)
    V = torch.randn(batch, T, Cin, generator=g) * spread
    Wt = torch.randn(batch, Cout, Cin, generator=g) * 0.1 / spread
    pk = _pack_b(Wt, Cin, Cout).to(dev)
    bias = torch.randn(Cout, generator=g) if extra else None
    res = torch.randn(batch, T, Cout, generator=g) if extra == 2 else None
    M = ops.gemm_h2p(V.to(dev), pk, batch, Cin, Cout, None if bias is None else bias.to(dev), None if res is None else res.to(dev)).cpu()
    ref = torch.einsum("btk,bok->bto", V.double(), Wt.double())
    if bias is not None:
        ref = ref + bias.double()
    if res is not None:
        ref = ref + res.double()
    e = rel_err(M, ref)
    print(f"gemm_h2p [{batch} x {T} x {Cin} x {Cout}], {decades} decades: rel err vs fp64 {e:.2e}")
    assert e < (3e-6 if decades <= 3 else 1e-4)


@pytest.mark.parametrize("kernel", [4, 5, 7])
@pytest.mark.parametrize("Cin", [16, 32, 48, 64, 80, 112, 128, 272])
def test_gemm_h2p_kernel_variants(dev, kernel, Cin):
    """Every tile shape of the fp16-pair kernel (library option bf3p_kernel: 4 = 256 x 256 tiles on the FOUR-SLOT RING -- two chunks per
    barrier: one to seventeen chunks, odd and even counts, prologue shorter than the ring --, 5 = 256 x 128, 7 = 128 x 128) against the
    fp64 product, with a ragged row count, bias and an in-place residual; the three shapes agree bit for bit (same terms, same order)."""
    import kernel_ops as ops
    from bbdm_amd import _lib
    g = torch.Generator().manual_seed(Cin)
    batch, T, Cout = 9, 300, 256
    V = torch.randn(batch, T, Cin, generator=g)
    Wt = torch.randn(batch, Cout, Cin, generator=g) * 0.1
    pk = _pack_b(Wt, Cin, Cout).to(dev)
    bias, res = torch.randn(Cout, generator=g), torch.randn(batch, T, Cout, generator=g)
    ref = torch.einsum("btk,bok->bto", V.double(), Wt.double()) + bias.double() + res.double()
    with _lib.option("bf3p_kernel", kernel):
        M = ops.gemm_h2p(V.to(dev), pk, batch, Cin, Cout, bias.to(dev), res.to(dev)).cpu()
    M0 = ops.gemm_h2p(V.to(dev), pk, batch, Cin, Cout, bias.to(dev), res.to(dev)).cpu()
    e = rel_err(M, ref)
    print(f"gemm_h2p kernel variant {kernel}, {Cin // 16} chunks: rel err vs fp64 {e:.2e}")
    assert e < 3e-6 and torch.equal(M, M0)


def test_gemm_h2p_bound_is_respected(dev):
    """The scale follows the BOUND, not the data: the same operands under bounds 1x ... 4096x their maximum give finite results whose
    error grows only once the second plane leaves the normal range, and a bound BELOW the maximum (a caller's bug) shows as inf / nan,
    never as a silently wrong finite number."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(1)
    batch, T, K, Cout = 2, 256, 256, 128
    V = torch.randn(batch, T, K, generator=g)
    Wt = torch.randn(batch, Cout, K, generator=g) * 0.05
    pk = _pack_b(Wt, K, Cout).to(dev)
    ref = torch.einsum("btk,bok->bto", V.double(), Wt.double())
    vmax = float(V.abs().max())
    errs = []
    for f in (1.0, 16.0, 512.0, 4096.0):
        M = ops.gemm_h2p(V.to(dev), pk, batch, K, Cout, bound_a=torch.full((1,), vmax * f, device=dev)).cpu()
        assert bool(torch.isfinite(M).all())
        errs.append(rel_err(M, ref))
    print("gemm_h2p error under bounds 1x / 16x / 512x / 4096x the maximum:", " ".join(f"{e:.2e}" for e in errs))
    assert errs[0] < 1e-6 and errs[1] < 1e-6 and errs[2] < 2e-6 and errs[3] < 2e-5
    M = ops.gemm_h2p(V.to(dev), pk, batch, K, Cout, bound_a=torch.full((1,), vmax / 64.0, device=dev)).cpu()
    assert not bool(torch.isfinite(M).all())


@pytest.mark.parametrize("batch,T,rows,Cin,Cout,splits", [(16, 256, 128, 1024, 1024, 4), (2, 256, 256, 256, 72, 2),
                                                          (3, 512, 288, 48, 260, 3), (16, 256, 32, 512, 128, 1),
                                                          (36, 256, 64, 2048, 512, 8)])
def test_gemm_bf3p_splitk(dev, batch, T, rows, Cin, Cout, splits):
    """The forward GEMM of the SMALL layers (csrc/gemm_bf3p.hip: bf3p_forward): only the row tiles holding real rows are computed,
    K is split and split z writes its partial sums to M[z]; the partials added in order are the fp32-accurate product (vs fp64), the
    rows beyond ``rows`` are never written, and one split is bit-equal to the unsplit kernel."""
    import kernel_ops as ops
    from bbdm_amd import _lib
    g = torch.Generator().manual_seed(Cin + Cout + rows)
    V = torch.randn(batch, T, Cin, generator=g) * torch.logspace(-2, 2, Cin)
    Wt = torch.randn(batch, Cout, Cin, generator=g) * 0.1 / torch.logspace(-2, 2, Cin)
    cout_pad = (Cout + 127) // 128 * 128
    pk = torch.zeros(batch, Cin // 16, cout_pad, 16)
    for c in range(Cin // 16):
        pk[:, c, :Cout, :] = Wt[:, :, c * 16:(c + 1) * 16]
    pk = pk.contiguous().to(dev)
    Mz = ops.gemm_bf3p_splitk(V.to(dev), pk, batch, Cin, Cout, rows, splits, fill=-7.0).cpu()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert bool((Mz[:, :, rows:] == -7.0).all())                      # rows beyond `rows`: untouched
    M = Mz[0, :, :rows].clone()
    for z in range(1, splits):
        M += Mz[z, :, :rows]
    ref = torch.einsum("btk,bok->bto", V[:, :rows].double(), Wt.double())
    e = rel_err(M, ref)
    print(f"gemm_bf3p split-K [{batch} x {rows}/{T} x {Cin} x {Cout}] / {splits}: rel err vs fp64 {e:.2e}")
    assert e < 3e-6
    if splits == 1:
        M0 = ops.gemm_bf3p(V.to(dev), pk, batch, Cin, Cout).cpu()
        assert torch.equal(M, M0[:, :rows])
    lib = _lib.load()
    s = lib.bbdm_gemm_bf3p_fwd_splits(batch, rows, Cin, Cout)
    assert 1 <= s <= max(1, Cin // 256) and lib.bbdm_gemm_bf3p_fwd_splits(64, 32768, 1024, 1024) == 1


@pytest.mark.parametrize("m,N,H,W,Cin,Cout,splits", [(2, 32, 4, 4, 1024, 128, 0), (2, 4, 8, 8, 512, 72, 2), (4, 3, 8, 16, 256, 136, 4),
                                                     (2, 2, 6, 10, 64, 40, 1)])
def test_winograd_splitk_stages(dev, m, N, H, W, Cin, Cout, splits):
    """A small 3x3 layer end to end: bf16-plane input transform -> split-K tile GEMMs -> output transform adding the partials, with
    bias, residual and the fused GroupNorm statistics, against the fp64 convolution.  splits = 0: the library's own choice."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(5 * m + H + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    ref = (F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()).float()
    ks = splits or lib.bbdm_winograd_gemm_bf3p_splits(m, N, H, W, Cin, Cout)
    if not splits:
        assert ks > 1, "the library splits this shape (32 images of 4x4: 16 workgroups of 128 x 128, K = 1024)"
    P, tiles = (m + 2) ** 2, lib.bbdm_winograd_tiles(m, N, H, W)
    st = ops._st(x.to(dev))
    xg, rg, bg = _nhwc(x).to(dev), _nhwc(r).to(dev), b.to(dev)
    pw = ops.pack_winograd_weight(w.to(dev), m=m)
    Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
    Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
    M = torch.full((ks * P * tiles * Cout,), float("nan"), device=dev)
    out = torch.empty(N, H, W, Cout, device=dev)
    cpg = Cout // 8 if Cout % 32 == 0 else 0
    stats = ops.new_stats(N, 32, dev)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, xg.data_ptr(), Cin, Vp.data_ptr(), None, None, 0, 0, 0, N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_bf3p_splitk_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, ks, st)
    _lib.call("bbdm_winograd_output_splitk_stats_f32", m, M.data_ptr(), bg.data_ptr(), rg.data_ptr(), Cout, out.data_ptr(), Cout, 0,
              N, H, W, Cout, stats.data_ptr() if cpg else None, cpg, 0, None, 0, 0, ks, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    o = _nchw(out.cpu())
    e = rel_err(o, ref)
    print(f"winograd split-K m={m} N{N} {H}x{W} {Cin}->{Cout} / {ks}: rel err {e:.2e}")
    assert e < WINO_TOL[m]
    if cpg:
        s_ref = o.double().reshape(N, 8, -1).sum(-1)
        assert float((ops.read_stats(stats).cpu()[:, :8, 0] - s_ref).abs().max()) < 1e-3 * max(1.0, float(s_ref.abs().max()))


@pytest.mark.parametrize("m,N,H,W,Cin,Cout,pre", [(6, 2, 12, 20, 32, 40, 1), (4, 1, 8, 16, 16, 24, 0), (2, 3, 4, 6, 16, 8, 1),
                                                  (6, 1, 7, 11, 48, 136, 0), (6, 2, 32, 32, 64, 128, 1),
                                                  # m = 7: the phase filters as 2 x 2 filters, F(7x7, 2x2) on the 8-point transform
                                                  (7, 2, 12, 20, 32, 128, 1), (7, 1, 7, 11, 48, 128, 0), (7, 2, 32, 32, 64, 256, 1),
                                                  (7, 1, 6, 13, 16, 128, 1), (7, 3, 14, 21, 16, 128, 0)])
def test_upsample_conv_as_phase_filters(dev, m, N, H, W, Cin, Cout, pre):
    """conv3x3(nearest x2 (act(x))) (Upsample.forward, openaimodel.py:111-121; the up-sampling ResBlock's first conv, :259-264) run as
    its four phase filters on the LOW-resolution x: bbdm_upsample_phase_weights_f32 -> Winograd with 4 Cout GEMM columns -> output
    transform scattering channel (2a + b) Cout + co of (i, j) to pixel (2i + a, 2j + b) (BBDM_CONV_OUT_PHASES), with the fused
    GroupNorm-coefficient producer and the consumer's GroupNorm statistics; against the fp64 convolution of the upsampled tensor."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11 * m + H + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    sc = torch.randn(N, Cin, generator=g)
    bi = torch.randn(N, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    a = x.double()
    if pre:
        a = a * sc.double()[:, :, None, None] + bi.double()[:, :, None, None]
        a = a * torch.sigmoid(a)
    ref = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1).float()
    st = ops._st(x.to(dev))
    wg = w.to(dev).contiguous()
    w4 = torch.empty(4 * Cout, Cin, 3, 3, device=dev)
    _lib.call("bbdm_upsample_phase_weights_f32", wg.data_ptr(), w4.data_ptr(), Cout, Cin, st)
    # the phase filters themselves: each is the 3x3 filter with its taps collapsed onto the 2 x-pixels they read
    w4c = w4.cpu().view(2, 2, Cout, Cin, 3, 3)
    assert torch.equal(w4c[0, 0, :, :, 0, 0], w[:, :, 0, 0]) and torch.equal(w4c[1, 1, :, :, 2, 2], w[:, :, 2, 2])
    assert torch.equal(w4c[0, 1, :, :, 1, 1], (w[:, :, 1, 0] + w[:, :, 2, 0]) + (w[:, :, 1, 1] + w[:, :, 2, 1]))
    assert float(w4c[0, :, :, :, 2].abs().max()) == 0.0 and float(w4c[:, 1, :, :, :, 0].abs().max()) == 0.0
    P, tiles = (64 if m == 7 else (m + 2) ** 2), lib.bbdm_winograd_tiles(m, N, H, W)
    pw = ops.pack_winograd_weight(w4, m=m)
    Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, 4 * Cout), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, 4 * Cout, st)
    Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
    M = torch.empty(P * tiles * 4 * Cout, device=dev)
    out = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device=dev)
    xg, scg, big, bg = _nhwc(x).to(dev), sc.to(dev), bi.to(dev), b.to(dev)
    cpg = Cout // 8 if Cout % 32 == 0 else 0
    stats = ops.new_stats(N, 32, dev)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, xg.data_ptr(), Cin, Vp.data_ptr(), scg.data_ptr() if pre else None,
              big.data_ptr() if pre else None, Cin, pre, 0, N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, 4 * Cout, st)
    _lib.call("bbdm_winograd_output_stats_f32", m, M.data_ptr(), bg.data_ptr(), None, 0, out.data_ptr(), Cout, 8, N, H, W, Cout,
              stats.data_ptr() if cpg else None, cpg, 0, None, 0, 0, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    o = _nchw(out.cpu())
    assert not bool(torch.isnan(o).any())                       # every pixel of the 2H x 2W result written
    e = rel_err(o, ref)
    print(f"upsample conv as phase filters m={m} N{N} {H}x{W}->{2 * H}x{2 * W} {Cin}->{Cout}: rel err {e:.2e}")
    assert e < WINO_TOL[m]
    if cpg:
        s_ref = o.double().reshape(N, 8, -1).sum(-1)
        assert float((ops.read_stats(stats).cpu()[:, :8, 0] - s_ref).abs().max()) < 1e-3 * max(1.0, float(s_ref.abs().max()))


@pytest.mark.parametrize("N,H,W,Cin,Cout,pre,up,res,f32v", [(2, 16, 24, 32, 128, 1, 0, 1, 0), (1, 13, 19, 64, 256, 1, 0, 0, 0),
                                                             (2, 16, 16, 32, 128, 1, 1, 2, 0), (3, 9, 8, 32, 128, 0, 0, 1, 1),
                                                             (1, 32, 32, 64, 128, 1, 0, 0, 1),
                                                             (2, 16, 24, 32, 128, 1, 0, 1, 2), (1, 13, 19, 64, 256, 1, 0, 0, 2),
                                                             (2, 16, 16, 32, 128, 1, 1, 2, 2), (3, 9, 8, 32, 128, 0, 0, 1, 2)])
def test_winograd_f8_forward(dev, N, H, W, Cin, Cout, pre, up, res, f32v):
    """F(8x8, 3x3) on ten points (round 5; forward only): the plane input transform (fused producer, nearest x2) -> pre-split tile
    GEMMs -> single-buffer two-phase output transform with bias, residual (full / per image), GroupNorm statistics and several tiles
    per workgroup; ragged images; ``f32v``: 1 = fp32 V rows + the GEMM that splits them, 2 = the fp16-pair planes (round 6: the bound of
    the transformed tensor = the maximum of the activated input).  Against the fp64 convolution."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    m = 8
    g = torch.Generator().manual_seed(8 * H + Cout + up)
    hs, ws_ = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(N, Cin, hs, ws_, generator=g)
    sc, bi = torch.randn(N, Cin, generator=g), torch.randn(N, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    a = x.double()
    if pre:
        a = F.silu(a * sc.double()[:, :, None, None] + bi.double()[:, :, None, None])
    if up:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    ref = F.conv2d(a, w.double(), b.double(), padding=1)
    r = None
    if res == 1:
        r = torch.randn(N, Cout, H, W, generator=g)
        ref = ref + r
        rg, flags = _nhwc(r).to(dev), 0
    elif res == 2:
        r = torch.randn(N, Cout, generator=g)
        ref = ref + r[:, :, None, None]
        rg, flags = r.to(dev), 2
    st = ops._st(x.to(dev))
    xg, scg, big, bg = _nhwc(x).to(dev), sc.to(dev), bi.to(dev), b.to(dev)
    P, tiles = 100, lib.bbdm_winograd_tiles(m, N, H, W)
    assert tiles % 256 == 0 and tiles >= N * -(-H // 8) * -(-W // 8)
    pw = ops.pack_winograd_weight(w.to(dev), m=m)
    M = torch.full((P * tiles * Cout,), float("nan"), device=dev)
    out = torch.full((N, H, W, Cout), float("nan"), device=dev)
    pre_args = (scg.data_ptr() if pre else None, big.data_ptr() if pre else None, Cin if pre else 0, pre)
    if f32v == 2:
        # the two-launch packing (fp32 G g G^T, then the planes under the tap bound x the gain of G . G^T) -- the fused launch
        # (bbdm_winograd_pack_weight_h2p_f32, what ops.conv3x3_winograd_planes and the product use) must write the same planes
        wd = w.to(dev).contiguous()
        pf = torch.empty(lib.bbdm_winograd_packed_floats(m, Cout, Cin), dtype=torch.float32, device=dev)
        _lib.call("bbdm_winograd_pack_weight_f32", m, wd.data_ptr(), pf.data_ptr(), Cout, Cin, Cin, 0, st)
        ub = ops.absmax(wd)
        Bp = torch.zeros(lib.bbdm_gemm_h2p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_gemm_h2p_pack_b_f32", pf.data_ptr(), Bp.data_ptr(), ub.data_ptr(), float(lib.bbdm_winograd_g_gain(m)), P, Cin, Cout, st)
        Bp2 = torch.zeros_like(Bp)
        _lib.call("bbdm_winograd_pack_weight_h2p_f32", m, wd.data_ptr(), Bp2.data_ptr(), Cout, Cin, Cin, 0, ub.data_ptr(), st)
        assert torch.equal(Bp.cpu(), Bp2.cpu())
        vb = torch.full((1,), float(a.abs().max()), dtype=torch.float32, device=dev)
        Vp = torch.empty(lib.bbdm_gemm_h2p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_winograd_input_h2p_f32", m, xg.data_ptr(), Cin, Vp.data_ptr(), *pre_args, up, N, H, W, Cin, vb.data_ptr(), st)
        _lib.call("bbdm_winograd_gemm_h2p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, vb.data_ptr(),
                  ub.data_ptr(), st)
    elif f32v:
        V = torch.empty(P * tiles * Cin, device=dev)
        _lib.call("bbdm_winograd_input_f32", m, xg.data_ptr(), Cin, V.data_ptr(), *pre_args, up, N, H, W, Cin, st)
        pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(P, Cin, Cout), dtype=torch.int16, device=dev)
        _lib.call("bbdm_gemm_bf3_pack_f32", pw.data_ptr(), pk.data_ptr(), P, Cin, Cout, st)
        _lib.call("bbdm_winograd_gemm_bf3_f32", m, V.data_ptr(), pk.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
    else:
        Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_winograd_pack_weight_bf3p_f32", m, w.to(dev).contiguous().data_ptr(), Bp.data_ptr(), Cout, Cin, Cin, 0, st)
        Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
        _lib.call("bbdm_winograd_input_bf3p_f32", m, xg.data_ptr(), Cin, Vp.data_ptr(), *pre_args, up, N, H, W, Cin, st)
        _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
    cpg = Cout // 32                                    # narrow groups: several tiles per workgroup (the single-buffer path's barrier)
    stats = ops.new_stats(N, 32, dev)
    _lib.call("bbdm_winograd_output_stats_f32", m, M.data_ptr(), bg.data_ptr(), rg.data_ptr() if res else None,
              Cout if res else 0, out.data_ptr(), Cout, flags if res else 0, N, H, W, Cout, stats.data_ptr(), cpg, 0, None, 0, 0, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    o = _nchw(out.cpu())
    assert not bool(torch.isnan(o).any())
    e = rel_err(o, ref.float())
    print(f"F(8x8,3x3) N{N} {H}x{W} {Cin}->{Cout} pre{pre} up{up} res{res} f32v{f32v}: rel err {e:.2e}")
    assert e < WINO_TOL[8]
    s_ref = o.double().reshape(N, 32, -1).sum(-1)
    assert float((ops.read_stats(stats).cpu()[:, :, 0] - s_ref).abs().max()) < 1e-3 * max(1.0, float(s_ref.abs().max()))


def _stress_layer(name, C, K, S, seed=0):
    """The weight / input sets of the round-5 verdict's probe: the m = 8 error is data-dependent (filters with a DC component, 3x gain,
    heavy tails, 30x spatial outliers), and a trained checkpoint is none of the zero-mean Gaussians the other tests draw."""
    g = torch.Generator().manual_seed(seed)
    x = F.silu(torch.randn(2, C, S, S, generator=g) * 1.5 + 0.3)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.02
    if name == "dc":
        w = w + 0.05
    elif name == "gain3":
        w = w * 3
    elif name == "student":
        torch.manual_seed(seed)
        w = torch.distributions.StudentT(3.0).sample((K, C, 3, 3)) * 0.02
    elif name == "outlier":
        for i, j in torch.randint(0, S, (40, 2), generator=g).tolist():
            x[:, :, i, j] *= 30
    elif name == "identity":
        w = w * 0.1
        for k in range(min(K, C)):
            w[k, k, 1, 1] += 1.0
    return x, w


@pytest.mark.parametrize("name", ["gauss", "dc", "gain3", "student", "outlier", "identity"])
@pytest.mark.parametrize("Cin", [128, 512])
def test_winograd_f8_stress_sets(dev, name, Cin):
    """One F(8x8, 3x3) layer on non-Gaussian data, both plane forms, against the fp64 convolution (round-5 verdict, item 1a): every set
    must stay below HALF the step's 1e-3 bar on the bf16x3 planes and below 2.5e-4 on the fp16-pair planes the product uses
    (measured MI355X: 0.2 - 1.5e-4; profiles/r06_parity_prints.txt)."""
    import kernel_ops as ops
    x, w = _stress_layer(name, Cin, 128, 32)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xg = _nhwc(x).to(dev)
    errs = {}
    for mode in ("bf3", "h2"):
        o = _nchw(ops.conv3x3_winograd_planes(xg, w.to(dev), None, 8, mode=mode).cpu())
        errs[mode] = rel_err(o, ref.float())
    print(f"F(8x8,3x3) stress set {name:8s} Cin={Cin}: max-norm error bf16x3 {errs['bf3']:.2e}  fp16-pair {errs['h2']:.2e}")
    assert errs["bf3"] < 5e-4 and errs["h2"] < 2.5e-4


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 24, 128, 96), (1, 13, 18, 256, 32)])
def test_winograd_f8_dgrad(dev, N, H, W, Cin, Cout):
    """F(8x8, 3x3) as the data-gradient convolution of a training plan (UNetModel.winograd_train8): the flipped, transposed filters
    straight into the bf16 planes (dgrad = 1), dY through the plane input transform, the pre-split tile GEMMs, the output transform
    without bias / statistics.  Against the fp64 transposed convolution; ragged image in the second case."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    m, P = 8, 100
    g = torch.Generator().manual_seed(8 * H + Cout)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    dy = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)                  # dX of conv2d(x, w, padding=1): [N, Cin, H, W]
    dyg, wg = _nhwc(dy).to(dev), w.to(dev).contiguous()
    st = ops._st(dyg)
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cout, Cin), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_winograd_pack_weight_bf3p_f32", m, wg.data_ptr(), Bp.data_ptr(), Cout, Cin, Cout, 1, st)
    Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cout), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, dyg.data_ptr(), Cout, Vp.data_ptr(), None, None, 0, 0, 0, N, H, W, Cout, st)
    M = torch.full((P * tiles * Cin,), float("nan"), device=dev)
    _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cout, Cin, st)
    out = torch.full((N, H, W, Cin), float("nan"), device=dev)
    _lib.call("bbdm_winograd_output_f32", m, M.data_ptr(), None, None, 0, out.data_ptr(), Cin, 0, N, H, W, Cin, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    o = _nchw(out.cpu())
    assert not bool(torch.isnan(o).any())
    e = rel_err(o, ref.float())
    print(f"F(8x8,3x3) data gradient N{N} {H}x{W} {Cin}<-{Cout}: rel err {e:.2e}")
    assert e < WINO_TOL[8]
    # ... and the fp32 packing of the same filters (the two-launch form) agrees with the fused planes
    a = ops.winograd_weight_planes(wg, m, Cout, True, fused=True)
    b = ops.winograd_weight_planes(wg, m, Cout, True, fused=False)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    value = lambda pl: (lambda h: (h[:, 0] + h[:, 1]) + h[:, 2])(pl.cpu().view(torch.int16).view(-1, 3, 512).view(torch.bfloat16).float())
    va, vb = value(a), value(b)
    assert torch.equal(va == 0, vb == 0) and float((va - vb).abs().max()) <= 2.0 ** -22 * float(vb.abs().max())


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", [(6, 2, 12, 20, 32, 40), (4, 1, 8, 16, 16, 24), (2, 3, 4, 6, 16, 8), (6, 1, 14, 10, 16, 72),
                                              (6, 2, 14, 10, 16, 128)])
def test_winograd_output_adds_upsampled_residual(dev, m, N, H, W, Cin, Cout):
    """BBDM_CONV_RES_UPSAMPLE: the output transform adds a residual given at half the resolution, nearest-upsampled x2 -- the
    skip path x_upd(x) of an up-sampling ResBlock (openaimodel.py:259-264) without a resampling pass."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(m + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H // 2, W // 2, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + F.interpolate(r.double(), scale_factor=2, mode="nearest")
    pw = ops.pack_winograd_weight(w.to(dev), m=m)
    out = ops.conv3x3_winograd(_nhwc(x).to(dev), pw, b.to(dev), Cout, residual=_nhwc(r).to(dev), m=m, res_upsample=True)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref.float()) < WINO_TOL[m]


@pytest.mark.parametrize("m,up,silu,N,H,W,Cin,Cout", [(6, 0, 1, 3, 16, 24, 64, 96), (6, 1, 1, 2, 20, 12, 32, 136),
                                                      (4, 0, 1, 3, 16, 24, 64, 96), (4, 1, 0, 1, 8, 8, 16, 8),
                                                      (2, 1, 1, 3, 16, 24, 64, 96), (6, 0, 0, 5, 7, 9, 48, 260)])
def test_winograd_bf3p_stages(dev, m, up, silu, N, H, W, Cin, Cout):
    """The Winograd path with the input transform writing the three bf16 planes (bbdm_winograd_input_bf3p_f32) and the tile GEMMs
    on the pre-split kernel: against the fp64 convolution, and against the bbdm_winograd_input_f32 + bbdm_winograd_gemm_bf3_f32
    pipeline (fp32 V, split while staged).  The two pipelines agree to the last few ulps of M, not bit for bit: the GEMMs are
    bit-equal on equal operands (test_gemm_bf3p_matches_bf3_bitwise), but the two input-transform kernels are compiled separately
    and hipcc contracts their multiply-adds into FMAs differently (bit-equal on the emulator, which builds with -ffp-contract=off)."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(31 + 2 * up + silu + m)
    hs, ws_ = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(N, Cin, hs, ws_, generator=g)
    sc = torch.randn(N, Cin, generator=g)
    bi = torch.randn(N, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    a = x.double() * sc.double()[:, :, None, None] + bi.double()[:, :, None, None]
    if silu:
        a = F.silu(a)
    if up:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    ref = F.conv2d(a, w.double(), b.double(), padding=1).float()
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    xg, scg, big, bg = _nhwc(x).to(dev), sc.to(dev), bi.to(dev), b.to(dev)
    pw = ops.pack_winograd_weight(w.to(dev), m=m)
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    P = (m + 2) ** 2
    Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
    Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
    M = torch.empty(P * tiles * Cout, device=dev)
    out = torch.empty(N, H, W, Cout, device=dev)
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, xg.data_ptr(), Cin, Vp.data_ptr(), scg.data_ptr(), big.data_ptr(), Cin, silu, up,
              N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
    _lib.call("bbdm_winograd_output_f32", m, M.data_ptr(), bg.data_ptr(), None, 0, out.data_ptr(), Cout, 0, N, H, W, Cout, st)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref) < WINO_TOL[m]
    # the fp32-V pipeline on gemm_bf3.hip
    V = torch.empty(P * tiles * Cin, device=dev)
    M0 = torch.empty(P * tiles * Cout, device=dev)
    pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(P, Cin, Cout), dtype=torch.int16, device=dev)
    _lib.call("bbdm_gemm_bf3_pack_f32", pw.data_ptr(), pk.data_ptr(), P, Cin, Cout, st)
    _lib.call("bbdm_winograd_input_f32", m, xg.data_ptr(), Cin, V.data_ptr(), scg.data_ptr(), big.data_ptr(), Cin, silu, up,
              N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_bf3_f32", m, V.data_ptr(), pk.data_ptr(), M0.data_ptr(), N, H, W, Cin, Cout, st)
    torch.cuda.synchronize()
    T_raw = N * -(-H // m) * -(-W // m)
    a_, b_ = M.view(P, tiles, Cout)[:, :T_raw].cpu(), M0.view(P, tiles, Cout)[:, :T_raw].cpu()
    assert rel_err(a_, b_) < 2e-6, rel_err(a_, b_)


@pytest.mark.parametrize("m,up,silu,N,H,W,Cin,Cout", [(6, 0, 1, 3, 16, 24, 64, 96), (6, 1, 1, 2, 20, 12, 32, 136), (4, 1, 0, 1, 8, 8, 16, 8),
                                                      (2, 1, 1, 3, 16, 24, 64, 96)])
def test_winograd_input_64bit_index_variant(dev, m, up, silu, N, H, W, Cin, Cout):
    """The input transform addresses its rows with 32-bit element indices and 24-bit multiplies; tensors of 2^32 elements (16 GB)
    or more take the IDX64 instantiation (csrc/winograd.hip: winograd_input_split2_kernel).  Option "wino_idx64" (read at every
    call) forces it on the ordinary test shapes."""
    from bbdm_amd import _lib
    with _lib.option("wino_idx64", 1):
        test_winograd_bf3p_stages(dev, m, up, silu, N, H, W, Cin, Cout)


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(256, 32, 40, False), (512, 64, 132, True), (1024, 1536, 512, True)])
def test_conv1x1_bf3(dev, pixels, Cin, Cout, res):
    """1x1 convolution / Linear on the bf16x3 kernel: bias, in-place residual, input taken from a channel slice of a wider
    buffer, output written into a slice."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(pixels + Cin)
    wide = torch.randn(pixels, Cin + 16, generator=g)
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(pixels, Cout, generator=g) if res else None
    ref = wide[:, 16:].double() @ w.double().t() + b.double() + (r.double() if res else 0)
    buf = r.clone().to(dev) if res else None
    out = ops.conv1x1_bf3(wide.to(dev), w.to(dev), b.to(dev), residual=buf, out=buf, cin=Cin, x_off=16)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 3e-6


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(256, 32, 40, False), (512, 64, 132, True), (1024, 1536, 512, True), (768, 48, 256, False),
                                                 (2080, 256, 1024, True), (96, 16, 8, False)])
def test_conv1x1_bf3q_bitwise(dev, pixels, Cin, Cout, res):
    """The same 1x1 convolution on the pipelined kernel whose waves split the fp32 A operand between their MFMAs
    (gemm_bf3q_pipe_kernel): bit-equal to bbdm_conv1x1_bf3_f32 where that kernel takes the shape (pixels % 256 == 0); ragged row
    tiles (pixels 2080, 96), 256- and 128-column tiles, K from one chunk up."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(pixels + Cin)
    wide = torch.randn(pixels, Cin + 16, generator=g)
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(pixels, Cout, generator=g) if res else None
    ref = wide[:, 16:].double() @ w.double().t() + b.double() + (r.double() if res else 0)
    buf = r.clone().to(dev) if res else None
    out = ops.conv1x1_bf3q(wide.to(dev), w.to(dev), b.to(dev), residual=buf, out=buf, cin=Cin, x_off=16).cpu()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert rel_err(out, ref) < 3e-6
    if pixels % 256 == 0:
        buf0 = r.clone().to(dev) if res else None
        out0 = ops.conv1x1_bf3(wide.to(dev), w.to(dev), b.to(dev), residual=buf0, out=buf0, cin=Cin, x_off=16).cpu()
        assert torch.equal(out, out0), (out - out0).abs().max()


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(256, 32, 40, False), (512, 64, 132, True), (1024, 1536, 512, True), (768, 48, 256, False),
                                                 (2080, 256, 1024, True), (96, 16, 8, False), (300, 112, 256, True)])
def test_conv1x1_h2q(dev, pixels, Cin, Cout, res):
    """The wide 1x1 convolutions on the fp16-pair planes (gemm_bf3q_pipe_kernel<NP = 2>: the waves scale and split the fp32 activation
    into two fp16 halves between their MFMAs; 256-row tiles, 256- and 128-column tiles, ragged rows, K from one chunk, bias + in-place
    residual, a channel slice of a wider buffer) against fp64, next to bbdm_conv1x1_bf3q_f32 on the same operands; and under a bound 512x
    the measured maximum -- what the statistics bound (bbdm_h2_stats_bound_f32) may hand it."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(pixels + Cin)
    wide = torch.randn(pixels, Cin + 16, generator=g)
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(pixels, Cout, generator=g) if res else None
    ref = wide[:, 16:].double() @ w.double().t() + b.double() + (r.double() if res else 0)
    buf = r.clone().to(dev) if res else None
    out = ops.conv1x1_h2q(wide.to(dev), w.to(dev), b.to(dev), residual=buf, out=buf, cin=Cin, x_off=16).cpu()
    buf3 = r.clone().to(dev) if res else None
    out3 = ops.conv1x1_bf3q(wide.to(dev), w.to(dev), b.to(dev), residual=buf3, out=buf3, cin=Cin, x_off=16).cpu()
    loose = torch.full((1,), 512.0 * float(wide.abs().max()), device=dev)
    buf2 = r.clone().to(dev) if res else None
    out2 = ops.conv1x1_h2q(wide.to(dev), w.to(dev), b.to(dev), residual=buf2, out=buf2, cin=Cin, x_off=16, xbound=loose).cpu()
    e, e3, e2 = rel_err(out, ref), rel_err(out3, ref), rel_err(out2, ref)
    print(f"conv1x1 [{pixels} x {Cin} -> {Cout}]: rel err vs fp64 h2 {e:.2e} (bound 512x: {e2:.2e}), bf16x3 {e3:.2e}")
    assert e < 3e-6 and e2 < 3e-6


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(512, 1024, 3072, False), (512, 64, 132, True), (96, 128, 8, False), (2080, 256, 1024, True),
                                                 (100, 192, 72, True), (64, 64, 64, False)])
def test_conv1x1_h2s(dev, pixels, Cin, Cout, res):
    """The small-problem 1x1 kernel on the fp16-pair planes (gemm_bf3s_kernel<NP = 2>: the shapes of test_conv1x1_bf3s_bitwise -- one step
    (K = 64) to sixteen, ragged row and column tiles, both tile orders, in-place residual): against fp64, and bit-equal to
    bbdm_conv1x1_h2q_f32 on the same operands (same two planes, same three terms in the same order)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(pixels + Cin)
    wide = torch.randn(pixels, Cin + 16, generator=g)
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(pixels, Cout, generator=g) if res else None
    ref = wide[:, 16:].double() @ w.double().t() + b.double() + (r.double() if res else 0)
    buf = r.clone().to(dev) if res else None
    out = ops.conv1x1_h2q(wide.to(dev), w.to(dev), b.to(dev), residual=buf, out=buf, cin=Cin, x_off=16, small=True).cpu()
    buf0 = r.clone().to(dev) if res else None
    out0 = ops.conv1x1_h2q(wide.to(dev), w.to(dev), b.to(dev), residual=buf0, out=buf0, cin=Cin, x_off=16).cpu()
    e = rel_err(out, ref)
    print(f"conv1x1 small [{pixels} x {Cin} -> {Cout}] on the fp16 pair: rel err vs fp64 {e:.2e}")
    assert e < 3e-6 and torch.equal(out, out0), (out - out0).abs().max()


def test_h2_stats_bound(dev):
    """bbdm_h2_stats_bound_f32: the largest root-sum-of-squares over the (image, group) cells of a GroupNorm accumulator bounds the raw
    tensor (also with one 1000x outlier), and is at most sqrt(values per group) above its maximum."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(3)
    N, H, W, C = 3, 12, 10, 64
    x = torch.randn(N, H, W, C, generator=g) * 2.5 + 0.7
    x[1, 3, 4, 17] = 3000.0
    xd = x.to(dev)
    stats = ops.groupnorm_stats(xd)
    bound = torch.zeros(1, device=dev)
    _lib.call("bbdm_h2_stats_bound_f32", stats.data_ptr(), N, 32, bound.data_ptr(), ops._st(xd))
    want = float(x.double().reshape(N, H * W, 32, C // 32).pow(2).sum((1, 3)).max().sqrt())
    assert float(x.abs().max()) <= float(bound) and abs(float(bound) - want) <= 1e-5 * want, (float(bound), want)


@pytest.mark.parametrize("m,up,silu,film,N,H,W,C", [(2, 0, 1, True, 3, 4, 4, 64), (4, 0, 1, True, 2, 16, 16, 128), (4, 1, 1, False, 2, 16, 8, 64),
                                                    (6, 0, 0, True, 2, 14, 20, 192), (2, 1, 1, True, 5, 8, 8, 256)])
def test_winograd_input_forms_groupnorm_coefficients_bitwise(dev, m, up, silu, film, N, H, W, C):
    """bbdm_winograd_input_bf3p_gn_f32 (the input transform forms sc / bi from the GroupNorm statistics, gamma, beta and the FiLM
    vector itself) writes the SAME planes, bit for bit, as bbdm_groupnorm_coeffs_f32 followed by bbdm_winograd_input_bf3p_f32 --
    with and without FiLM / SiLU / nearest upsampling, F(2) / F(4) / F(6), ragged tile grids."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(7 * m + up + 3 * silu + C)
    hs, ws_ = (H // 2, W // 2) if up else (H, W)
    x = (torch.randn(N, hs, ws_, C, generator=g) * 1.7 + 0.3).to(dev)
    gamma, beta = (torch.randn(C, generator=g) * 0.5 + 1).to(dev), torch.randn(C, generator=g).to(dev)
    fv = torch.randn(N, 2 * C + 8, generator=g).to(dev) * 0.3 if film else None
    lib = _lib.load()
    st = ops._st(x)
    stats = ops.groupnorm_stats(x)
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    P = (m + 2) ** 2
    nbytes = lib.bbdm_gemm_bf3p_a_bytes(P, tiles, C)
    sc = torch.empty(N, C, device=dev)
    bi = torch.empty(N, C, device=dev)
    _lib.call("bbdm_groupnorm_coeffs_f32", stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None if fv is None else fv.data_ptr(),
              0 if fv is None else fv.shape[1], sc.data_ptr(), bi.data_ptr(), C, N, hs * ws_, C, 32, 1e-5, st)
    V0 = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    V1 = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), C, V0.data_ptr(), sc.data_ptr(), bi.data_ptr(), C, silu, up, N, H, W, C, st)
    _lib.call("bbdm_winograd_input_bf3p_gn_f32", m, x.data_ptr(), C, V1.data_ptr(), stats.data_ptr(), None, C, silu, up, N, H, W, C,
              gamma.data_ptr(), beta.data_ptr(), None if fv is None else fv.data_ptr(), 0 if fv is None else fv.shape[1], hs * ws_, 32,
              1e-5, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert V0.any() and torch.equal(V0.cpu(), V1.cpu())


@pytest.mark.parametrize("m,up,silu,film,N,H,W,C", [(2, 0, 1, True, 3, 4, 4, 64), (4, 0, 1, True, 2, 16, 16, 128), (4, 1, 1, False, 2, 16, 8, 64),
                                                    (6, 0, 0, True, 2, 14, 20, 192), (8, 0, 1, True, 2, 16, 24, 64)])
def test_h2_bounds_and_planes_of_a_groupnorm_fed_layer(dev, m, up, silu, film, N, H, W, C):
    """The fp16-pair input transform behind a GroupNorm (round 6): (1) bbdm_h2_gn_bounds_f32 -- the bound from gamma / beta / the FiLM
    vector alone -- is >= the largest activated value whatever the data (here with 50x outliers in x) and within sqrt(n_g) of it; (2) the
    planes the transform writes under that bound reproduce B^T d B of the activated tensor to 2^-22 of the LAYER's largest entry (two
    11-bit planes); (3) the coefficient-folding variant writes the same planes bit for bit (m <= 6)."""
    import struct
    from bbdm_amd import _lib
    import kernel_ops as ops
    from test_winograd_math_cpu import MATS
    g = torch.Generator().manual_seed(7 * m + up + 3 * silu + C)
    hs, ws_ = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(N, hs, ws_, C, generator=g) * 1.7 + 0.3
    x[0, 1, 2, :] *= 50.0
    x = x.to(dev)
    gamma, beta = (torch.randn(C, generator=g) * 0.5 + 1).to(dev), torch.randn(C, generator=g).to(dev)
    fv = torch.randn(N, 2 * C + 8, generator=g).to(dev) * 0.3 if film else None
    lib = _lib.load()
    st = ops._st(x)
    stats = ops.groupnorm_stats(x)
    sc, bi = torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
    _lib.call("bbdm_groupnorm_coeffs_f32", stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None if fv is None else fv.data_ptr(),
              0 if fv is None else fv.shape[1], sc.data_ptr(), bi.data_ptr(), C, N, hs * ws_, C, 32, 1e-5, st)
    # (1) the bound
    n_g = hs * ws_ * (C // 32)
    table = torch.frombuffer(bytearray(struct.pack("QQiiff", gamma.data_ptr(), beta.data_ptr(), 0 if film else -1, C, (n_g - 1) ** 0.5, 1.0)),
                             dtype=torch.uint8).to(dev)
    bound = torch.zeros(1, device=dev)
    _lib.call("bbdm_h2_gn_bounds_f32", table.data_ptr(), 1, None if fv is None else fv.data_ptr(), 0 if fv is None else fv.shape[1], N,
              bound.data_ptr(), st)
    act = x.double().cpu() * sc.double().cpu()[:, None, None, :] + bi.double().cpu()[:, None, None, :]
    if silu:
        act = F.silu(act)
    amax = float(act.abs().max())
    print(f"h2 bound m={m} C={C}: bound {float(bound):.3g}, largest activated value {amax:.3g}, sqrt(n_g) {n_g ** 0.5:.1f}")
    assert amax <= float(bound) <= 4.0 * (n_g ** 0.5) * max(amax, 1.0)
    # (2) the planes
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    P = (m + 2) ** 2
    nbytes = lib.bbdm_gemm_h2p_a_bytes(P, tiles, C)
    V0 = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("bbdm_winograd_input_h2p_f32", m, x.data_ptr(), C, V0.data_ptr(), sc.data_ptr(), bi.data_ptr(), C, silu, up, N, H, W, C,
              bound.data_ptr(), st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    e = 14 - (int(torch.tensor(float(bound) * float(lib.bbdm_winograd_input_gain(m))).view(torch.int32) >> 23 & 0xff) - 127)
    pl = V0.cpu().view(torch.float16).double().reshape(P, tiles // 32, C // 16, 2, 2, 32, 8)       # [xi][rg][chunk][plane][k half][row][k % 8]
    V = (pl[:, :, :, 0] + pl[:, :, :, 1]).permute(0, 1, 4, 2, 3, 5).reshape(P, tiles, C) * 2.0 ** -e     # -> [xi][tile][channel]
    a_nchw = act.permute(0, 3, 1, 2)
    if up:
        a_nchw = F.interpolate(a_nchw, scale_factor=2, mode="nearest")
    BT = MATS[m][0]
    th, tw = -(-H // m), -(-W // m)
    pad = F.pad(a_nchw, (1, m * tw + 1 - W, 1, m * th + 1 - H))
    win = pad.unfold(2, m + 2, m).unfold(3, m + 2, m)                                              # N, C, th, tw, a, a
    want = torch.einsum("ij,nctwjk,lk->ilntwc", BT, win, BT).reshape(P, N * th * tw, C)
    err = float((V[:, :N * th * tw] - want).abs().max() / want.abs().max())
    print(f"   planes vs B^T d B: max error / max |V| = {err:.2e}")
    assert err < 2.0 ** -21 and float(V[:, N * th * tw:].abs().max() if tiles > N * th * tw else 0.0) == 0.0
    # (3) the coefficient-folding variant
    if m <= 6:
        V1 = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        _lib.call("bbdm_winograd_input_h2p_gn_f32", m, x.data_ptr(), C, V1.data_ptr(), stats.data_ptr(), None, C, silu, up, N, H, W, C,
                  gamma.data_ptr(), beta.data_ptr(), None if fv is None else fv.data_ptr(), 0 if fv is None else fv.shape[1], hs * ws_, 32,
                  1e-5, bound.data_ptr(), st)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        assert V0.any() and torch.equal(V0.cpu(), V1.cpu())


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(512, 1024, 3072, False), (512, 64, 132, True), (96, 128, 8, False), (2080, 256, 1024, True),
                                                 (100, 192, 72, True), (64, 64, 64, False)])
def test_conv1x1_bf3s_bitwise(dev, pixels, Cin, Cout, res):
    """The small-problem 1x1 kernel (gemm_bf3s_kernel: 64 x 64 tiles, 64 channels per step, everything by LDS-DMA two steps ahead,
    the fp32 operand split at the fragment read): bit-equal to bbdm_conv1x1_bf3q_f32 -- one step (K = 64) to sixteen, ragged row and
    column tiles, the XCD-owned tile order (Cout / 64 a multiple of 8) and the plain one, in-place residual."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(pixels + Cin)
    wide = torch.randn(pixels, Cin + 16, generator=g)
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    r = torch.randn(pixels, Cout, generator=g) if res else None
    ref = wide[:, 16:].double() @ w.double().t() + b.double() + (r.double() if res else 0)
    buf = r.clone().to(dev) if res else None
    out = ops.conv1x1_bf3q(wide.to(dev), w.to(dev), b.to(dev), residual=buf, out=buf, cin=Cin, x_off=16, small=True).cpu()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert rel_err(out, ref) < 3e-6
    buf0 = r.clone().to(dev) if res else None
    out0 = ops.conv1x1_bf3q(wide.to(dev), w.to(dev), b.to(dev), residual=buf0, out=buf0, cin=Cin, x_off=16).cpu()
    assert torch.equal(out, out0), (out - out0).abs().max()


def _group_sums(y_nhwc, cpg, coff, ctot_groups=32):
    """fp64 [N][32][2] (sum, sum of squares) of the channels of y placed at offset coff in a tensor with groups of cpg."""
    N, H, W, C = y_nhwc.shape
    out = torch.zeros(N, ctot_groups, 2, dtype=torch.float64)
    yd = y_nhwc.double()
    for c in range(C):
        g = (coff + c) // cpg
        out[:, g, 0] += yd[..., c].sum(dim=(1, 2))
        out[:, g, 1] += (yd[..., c] ** 2).sum(dim=(1, 2))
    return out


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", [(6, 2, 14, 20, 32, 64), (4, 3, 8, 12, 16, 128), (2, 5, 4, 4, 16, 32), (6, 9, 6, 6, 16, 32),
                                              (6, 2, 14, 20, 16, 128), (6, 3, 7, 9, 16, 256), (6, 2, 64, 64, 64, 512)])
def test_winograd_output_accumulates_groupnorm_statistics(dev, m, N, H, W, Cin, Cout):
    """bbdm_winograd_output_stats_f32: the output transform also accumulates the (image, group) sums two GroupNorm consumers
    need -- the next block's norm over [Cout] and an output block's norm over a concat [Cout' + Cout] -- so that no separate
    statistics pass re-reads the tensor.  The last case spans more images per workgroup than the LDS table holds."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(m * 100 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, H, W, Cout, generator=g)
    xg = ops.nchw_to_nhwc(x.to(dev), cpad=Cin)
    pw = ops.pack_winograd_weight(w.to(dev), in_pad=Cin, m=m)
    st = None if dev.type != "cuda" else torch.cuda.current_stream().cuda_stream
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    P = (m + 2) ** 2
    V = torch.zeros(P * tiles * Cin, device=dev)
    M = torch.zeros(P * tiles * Cout, device=dev)
    out = r.clone().to(dev)                                       # in-place residual
    cpg0, coff1 = Cout // 32 * 4 if Cout >= 128 else 4, 64                   # consumer 1: a [64 + Cout]-channel concat
    cpg1 = max(8, (-(-(coff1 + Cout) // 32) + 3) // 4 * 4)
    s0 = ops.new_stats(N, 32, dev)
    s1 = ops.new_stats(N, 32, dev)
    _lib.call("bbdm_winograd_input_f32", m, xg.data_ptr(), Cin, V.data_ptr(), None, None, 0, 0, 0, N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_f32", m, V.data_ptr(), pw.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
    _lib.call("bbdm_winograd_output_stats_f32", m, M.data_ptr(), b.to(dev).data_ptr(), out.data_ptr(), Cout, out.data_ptr(),
              Cout, 0, N, H, W, Cout, s0.data_ptr(), cpg0, 0, s1.data_ptr(), cpg1, coff1, st)
    torch.cuda.synchronize()
    y = out.cpu()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + r.double()
    assert rel_err(y, ref.float()) < WINO_TOL[m]
    for got, cpg, coff in ((s0, cpg0, 0), (s1, cpg1, coff1)):
        want = _group_sums(y, cpg, coff)
        assert float((ops.read_stats(got).cpu() - want).abs().max()) < 1e-9 * max(1.0, float(want.abs().max()))


def test_conv2d_accumulates_groupnorm_statistics(dev):
    """bbdm_conv2d_nhwc_stats_f32 (direct kernel epilogue) on a shape bbdm_conv_stats_fusable() accepts, and its refusal
    where a workgroup tile spans several images."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    N, H, W, Cin, Cout = 3, 16, 16, 32, 64
    assert lib.bbdm_conv_stats_fusable(N, H, W, Cin, Cout, 3) == 1 and lib.bbdm_conv_stats_fusable(8, 4, 4, Cin, Cout, 3) == 0
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xg = ops.nchw_to_nhwc(x.to(dev), cpad=Cin)
    pk = ops.pack_conv_weight(w.to(dev), cin_pad=Cin)
    out = torch.empty(N, H, W, Cout, device=dev)
    s0 = ops.new_stats(N, 32, dev)
    s1 = ops.new_stats(N, 32, dev)
    st = None if dev.type != "cuda" else torch.cuda.current_stream().cuda_stream
    _lib.call("bbdm_conv2d_nhwc_stats_f32", xg.data_ptr(), Cin, pk.data_ptr(), b.to(dev).data_ptr(), None, 0, out.data_ptr(),
              Cout, 0, None, 0, None, None, 0, 0, N, H, W, Cin, Cout, 3, s0.data_ptr(), 2, 0, s1.data_ptr(), 4, 32, st)
    torch.cuda.synchronize()
    y = out.cpu()
    assert rel_err(_nchw(y), F.conv2d(x, w, b, padding=1)) < TOL
    for got, cpg, coff in ((s0, 2, 0), (s1, 4, 32)):
        want = _group_sums(y, cpg, coff)
        assert float((ops.read_stats(got).cpu() - want).abs().max()) < 1e-9 * max(1.0, float(want.abs().max()))
    with pytest.raises(_lib.BBDMHipError, match="statistics"):
        _lib.call("bbdm_conv2d_nhwc_stats_f32", xg.data_ptr(), Cin, pk.data_ptr(), None, None, 0, out.data_ptr(), Cout, 0, None,
                  0, None, None, 0, 0, 8, 4, 4, Cin, Cout, 3, s0.data_ptr(), 2, 0, None, 0, 0, st)


@pytest.mark.parametrize("Cin", [8, 4])
def test_stem_conv_accumulates_groupnorm_statistics(dev, Cin):
    """The stem kernel (8 or 4 padded input channels -> 128, conv3x3_stem_kernel) with the GroupNorm statistics of its output for two
    consumers, ragged tiles, more tiles than one workgroup round would need per image."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    N, H, W, Cout = 2, 40, 56, 128
    assert lib.bbdm_conv_stats_fusable(N, H, W, Cin, Cout, 3) == 1
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xg = ops.nchw_to_nhwc(x.to(dev), cpad=Cin)
    pk = ops.pack_conv_weight(w.to(dev), cin_pad=Cin)
    out = torch.empty(N, H, W, Cout, device=dev)
    s0 = ops.new_stats(N, 32, dev)
    s1 = ops.new_stats(N, 32, dev)
    st = None if dev.type != "cuda" else torch.cuda.current_stream().cuda_stream
    _lib.call("bbdm_conv2d_nhwc_stats_f32", xg.data_ptr(), Cin, pk.data_ptr(), b.to(dev).data_ptr(), None, 0, out.data_ptr(),
              Cout, 0, None, 0, None, None, 0, 0, N, H, W, Cin, Cout, 3, s0.data_ptr(), 4, 0, s1.data_ptr(), 8, 64, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    y = out.cpu()
    assert rel_err(_nchw(y), F.conv2d(x, w, b, padding=1)) < TOL
    for got, cpg, coff in ((s0, 4, 0), (s1, 8, 64)):
        want = _group_sums(y, cpg, coff)
        assert float((ops.read_stats(got).cpu() - want).abs().max()) < 1e-9 * max(1.0, float(want.abs().max()))


def test_conv3x3_winograd_rejects_bad_shapes(dev):
    from bbdm_amd import _lib
    import kernel_ops as ops
    pw = ops.pack_winograd_weight(torch.zeros(16, 16, 3, 3, device=dev))
    with pytest.raises(_lib.BBDMHipError, match="multiples of m"):
        ops.conv3x3_winograd(torch.zeros(1, 5, 4, 16, device=dev), pw, None, 16)
    with pytest.raises(_lib.BBDMHipError, match="multiples of m"):
        ops.conv3x3_winograd(torch.zeros(1, 6, 8, 16, device=dev), pw, None, 16, m=4)
    with pytest.raises(_lib.BBDMHipError, match="unsupported"):
        ops.conv3x3_winograd(torch.zeros(1, 6, 6, 16, device=dev), pw, None, 16, m=3)


def test_conv2d_channel_slices(dev):
    """Reading from / writing into channel slices of wider buffers (the copy-free th.cat)."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 16, 16, 32, 128
    wide_in = torch.randn(N, H, W, 80, generator=g).to(dev)       # x lives in channels [48, 80)
    wide_out = torch.zeros(N, H, W, 320, device=dev)              # out goes to channels [64, 192)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    pw = ops.pack_conv_weight(w.to(dev))
    st = torch.cuda.current_stream().cuda_stream
    bg = b.to(dev)
    _lib.call("bbdm_conv2d_nhwc_f32", wide_in.data_ptr() + 4 * 48, 80, pw.data_ptr(), bg.data_ptr(), None, 0,
              wide_out.data_ptr() + 4 * 64, 320, 0, None, 0, None, None, 0, 0, N, H, W, Cin, Cout, 3, st)
    torch.cuda.synchronize()
    ref = F.conv2d(_nchw(wide_in.cpu()[..., 48:80]), w, b, padding=1)
    got = wide_out.cpu()
    assert rel_err(_nchw(got[..., 64:192]), ref) < TOL
    assert float(got[..., :64].abs().max()) == 0 and float(got[..., 192:].abs().max()) == 0


def test_conv2d_linearity_at_full_size(dev):
    """BASELINE config-2 sized layer (N=16, 64x64, 1024->1024): size-independent property instead of a CPU conv:
    conv(a x1 + b x2) == a conv(x1) + b conv(x2) (bias-free), plus a spot check of 64 output pixels on the CPU."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 16, 64, 64, 1024
    x1 = torch.randn(N, H, W, C, generator=g).to(dev)
    x2 = torch.randn(N, H, W, C, generator=g).to(dev)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.02
    pw = ops.pack_conv_weight(w.to(dev))
    y1 = ops.conv2d_nhwc(x1, pw, None, C, 3)
    y2 = ops.conv2d_nhwc(x2, pw, None, C, 3)
    y12 = ops.conv2d_nhwc(0.5 * x1 - 2.0 * x2, pw, None, C, 3)
    torch.cuda.synchronize()
    err = float(((0.5 * y1 - 2.0 * y2) - y12).abs().max() / y12.abs().max())
    assert err < 1e-5
    # spot check: one 3x3-neighbourhood dot product per sampled output, in float64 on the CPU
    xs = x1.cpu()
    idx = torch.randint(0, N * H * W, (64,), generator=g)
    for i in idx.tolist():
        n, h, ww = i // (H * W), (i // W) % H, i % W
        acc = torch.zeros(C, dtype=torch.float64)
        for r in range(3):
            for s in range(3):
                hh, wq = h + r - 1, ww + s - 1
                if 0 <= hh < H and 0 <= wq < W:
                    acc += w[:, :, r, s].double() @ xs[n, hh, wq].double()
        got = y1[n, h, ww].cpu().double()
        assert float((got - acc).abs().max() / acc.abs().max()) < 2e-5


GN_CASES = [(2, 8, 8, 96), (1, 4, 4, 192), (2, 16, 16, 128), (1, 8, 8, 640), (3, 4, 4, 1536), (2, 6, 10, 2048), (1, 32, 32, 32), (4, 64, 64, 256)]


@pytest.mark.parametrize("N,H,W,C", GN_CASES)
@pytest.mark.parametrize("mode", ["plain", "film_silu", "silu_pool", "silu_up"])
def test_groupnorm(dev, N, H, W, C, mode):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 2.0 + 0.7
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    film = 0.3 * torch.randn(N, 2 * C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, 1e-5)
    xg = _nhwc(x).to(dev)
    stats = ops.groupnorm_stats(xg)
    if mode == "plain":
        out = ops.groupnorm_apply(xg, stats, gamma.to(dev), beta.to(dev))
    elif mode == "film_silu":
        sc, sh = film[:, :C, None, None], film[:, C:, None, None]
        ref = F.silu(ref * (1 + sc) + sh)
        out = ops.groupnorm_apply(xg, stats, gamma.to(dev), beta.to(dev), film=film.to(dev), silu=True)
    elif mode == "silu_pool":
        if H % 2 or W % 2:
            pytest.skip("odd size")
        ref = F.avg_pool2d(F.silu(ref), 2, 2)
        out = ops.groupnorm_apply(xg, stats, gamma.to(dev), beta.to(dev), silu=True, resample=1)
    else:
        ref = F.interpolate(F.silu(ref), scale_factor=2, mode="nearest")
        out = ops.groupnorm_apply(xg, stats, gamma.to(dev), beta.to(dev), silu=True, resample=2)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref) < TOL


def test_resample_only(dev):
    import kernel_ops as ops
    x = torch.randn(2, 64, 8, 12)
    xg = _nhwc(x).to(dev)
    down = ops.groupnorm_apply(xg, None, None, None, resample=1)
    up = ops.groupnorm_apply(xg, None, None, None, resample=2)
    torch.cuda.synchronize()
    assert rel_err(_nchw(down.cpu()), F.avg_pool2d(x, 2, 2)) < 1e-6
    assert torch.equal(_nchw(up.cpu()), F.interpolate(x, scale_factor=2, mode="nearest"))


ATTN_CASES = [(2, 16, 4, 64), (1, 256, 2, 64), (2, 100, 3, 32), (3, 16, 2, 16), (1, 1024, 16, 64), (2, 37, 1, 64)]


@pytest.mark.parametrize("N,T,heads,ch", ATTN_CASES)
@pytest.mark.parametrize("new_order", [False, True])
def test_attention(dev, N, T, heads, ch, new_order):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(T + heads)
    C = heads * ch
    qkv = torch.randn(N, 3 * C, T, generator=g) * 1.5
    # reference: QKVAttentionLegacy / QKVAttention (oracle restatement)
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(N * heads, ch, T) for z in (q, k, v))
    else:
        q, k, v = qkv.reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v).reshape(N, C, T)
    out = ops.attention(qkv.permute(0, 2, 1).contiguous().to(dev), heads, new_order)
    torch.cuda.synchronize()
    assert rel_err(out.cpu().permute(0, 2, 1), ref) < TOL


@pytest.mark.parametrize("N,Tq,Tk,heads,ch", [(2, 64, 64, 2, 16), (1, 16, 64, 4, 16), (2, 100, 37, 3, 32), (1, 256, 1024, 2, 64)])
def test_cross_attention(dev, N, Tq, Tk, heads, ch):
    """CrossAttention.forward (attention.py:170-194) with its own key / value source (Tk != Tq)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(Tq + Tk)
    C = heads * ch
    q, k, v = (torch.randn(N, T, C, generator=g) for T in (Tq, Tk, Tk))
    sp = lambda t: t.reshape(N, t.shape[1], heads, ch).permute(0, 2, 1, 3).double()
    sim = torch.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * ch ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(N, Tq, C)
    out = ops.cross_attention(q.to(dev), k.to(dev), v.to(dev), heads)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < TOL


@pytest.mark.parametrize("rows,C", [(7, 64), (130, 32), (64, 1024), (3, 260)])
def test_layernorm_and_geglu(dev, rows, C):
    """nn.LayerNorm over the token's channels and GEGLU (attention.py:38-46,204-206)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 3 + 0.7
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = ops.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5)
    a = torch.randn(rows, 2 * C, generator=g) * 2
    z = ops.geglu(a.to(dev))
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5)) < TOL
    xa, gate = a.double().chunk(2, dim=-1)
    assert rel_err(z.cpu(), xa * F.gelu(gate)) < TOL


def test_attention_forces_rescale(dev):
    """A key whose score dwarfs all earlier ones appears late: the online-softmax rescale branch must be exact."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(3)
    N, T, heads, ch = 1, 160, 1, 64
    qkv = torch.randn(N, 3 * ch, T, generator=g)
    qkv[0, ch:2 * ch, 150] = qkv[0, :ch, 7] * 6.0          # k_150 aligned with q_7 -> huge score in the last tile
    q, k, v = qkv.reshape(1, 3 * ch, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", (q * s).double(), (k * s).double()), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v.double()).float()
    out = ops.attention(qkv.permute(0, 2, 1).contiguous().to(dev), heads)
    torch.cuda.synchronize()
    assert rel_err(out.cpu().permute(0, 2, 1), ref) < TOL


@pytest.mark.parametrize("kernel", [4, 5, 6])
def test_gemm_bf3p_ragged_rows_read_the_padding(dev, kernel):
    """A ragged last row tile of the pre-split GEMM (rows % 256 != 0): option "bf3p_pad_rows" decides what its idle 32-row blocks
    multiply -- the producer's zero rows behind the real ones (1, the default: the kernel is bound by the power its operand data draws)
    or a re-read of the last real rows (0) -- and must not change a stored value.  The Winograd path at 12 x 20 pixels, batch 5: 40 tiles
    of F(6x6) = 64 real rows of a 256-row layout."""
    from bbdm_amd import _lib
    with _lib.option("bf3p_kernel", kernel):
        outs = []
        for pad in (1, 0):
            with _lib.option("bf3p_pad_rows", pad):
                outs.append(_winograd_bf3p_M(dev, 6, 5, 12, 20, 32, 136))
    T = 5 * 2 * 4
    a, b = (o.view(64, -1, 136)[:, :T] for o in outs)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())


def _winograd_bf3p_M(dev, m, N, H, W, Cin, Cout):
    from bbdm_amd import _lib
    import kernel_ops as ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(H + W + Cin)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev)
    st = ops._st(x)
    P, tiles = (m + 2) ** 2, lib.bbdm_winograd_tiles(m, N, H, W)
    pw = ops.pack_winograd_weight(w, m=m)
    Bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(P, Cin, Cout), dtype=torch.uint8, device=dev)
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", pw.data_ptr(), Bp.data_ptr(), P, Cin, Cout, st)
    Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(P, tiles, Cin), dtype=torch.uint8, device=dev)
    M = torch.zeros(P * tiles * Cout, device=dev)
    _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), Cin, Vp.data_ptr(), None, None, 0, 0, 0, N, H, W, Cin, st)
    _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), Bp.data_ptr(), M.data_ptr(), N, H, W, Cin, Cout, st)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return M


@pytest.mark.parametrize("N,Tq,Tk,heads,ch", [(2, 160, 160, 2, 64), (1, 100, 37, 3, 32), (1, 256, 1000, 2, 64), (2, 33, 32, 1, 32),
                                              (1, 64, 31, 2, 64)])
def test_attention_interleaved_loop_is_bit_equal(dev, N, Tq, Tk, heads, ch):
    """The attention forward deals its operand splits between the MFMAs (option "attn_pipe" = 1, the default) instead of running them
    in phases of their own (0): the same arithmetic in another issue order -- same bits, incl. a single key tile, a ragged last tile
    and the log-sum-exp the training backward reads; and option "attn_bf3" = 0 / 2 (f32 MFMA / only Q K^T on the bf16x3 path) stay
    within the fp32 tolerance of it."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(Tq + Tk + ch)
    C = heads * ch
    q, k, v = (torch.randn(N, T, C, generator=g).to(dev) for T in (Tq, Tk, Tk))
    with _lib.option("attn_pipe", 1):
        a, la = ops.cross_attention(q, k, v, heads, return_lse=True)
    with _lib.option("attn_pipe", 0):
        b, lb = ops.cross_attention(q, k, v, heads, return_lse=True)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(la, lb)
    for mode in (0, 2):
        with _lib.option("attn_bf3", mode):
            c = ops.cross_attention(q, k, v, heads)
        assert rel_err(c.cpu(), a.cpu()) < TOL, mode


@pytest.mark.parametrize("N,T,heads,ch,new_order", [(2, 128, 2, 64, False), (1, 256, 3, 32, True), (2, 384, 1, 64, True), (1, 1024, 2, 64, False)])
def test_attention_presplit_form_is_bit_equal(dev, N, T, heads, ch, new_order):
    """bbdm_attention_kv_planes_f32 + bbdm_attention_planes_f32 (K / V split into their bf16 operand planes once per head, copied into
    LDS by LDS-DMA) against bbdm_attention_f32 (split by every workgroup): the same split, the same MFMA order -- the same bits, output
    and log-sum-exp; one, two and many key tiles per query block, both channel orders; shapes the form does not take report 0 bytes."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(T + ch)
    qkv = torch.randn(N, T, 3 * heads * ch, generator=g).to(dev)
    a, la = ops.attention(qkv, heads, new_order, return_lse=True)
    with _lib.option("attn_pipe", 3):
        b, lb = ops.attention_planes(qkv, heads, new_order, return_lse=True)
        lib = _lib.load()
        assert lib.bbdm_attention_kv_planes_bytes(N, T + 32, heads, ch) == 0 and lib.bbdm_attention_kv_planes_bytes(N, T, heads, 16) == 0
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(la, lb)
    with _lib.option("attn_pipe", 1):
        assert ops.attention_planes(qkv, heads, new_order) is None
    if T < 1024:
        assert ops.attention_planes(qkv, heads, new_order) is None          # default setting: long sequences only


@pytest.mark.parametrize("N,T,heads,ch,new_order,slack", [(2, 128, 2, 64, False, 1.0), (1, 256, 3, 32, True, 1.0), (1, 1024, 2, 64, False, 1.0),
                                                          (2, 384, 1, 64, True, 4096.0), (1, 256, 2, 32, False, 2048.0), (1, 4096, 2, 64, False, 4096.0)])     # (the last: C2's sequence length)
def test_attention_h2(dev, N, T, heads, ch, new_order, slack):
    """The pre-split attention on the fp16-pair planes (bbdm_attention_kv_planes_h2_f32 + bbdm_attention_planes_h2_f32: q, k, v under one
    power-of-two scale from a bound of the qkv tensor, the softmax weights under their exact bound 1, three f16 MFMA terms per product)
    against an fp64 attention: within the bar, and no worse than the six-term bf16x3 form on the same input -- also under a bound far
    above the data (the provable bound of a qkv projection sits ~2^13 above typical values; here up to 4096 x an outlier 7 x the typical
    value -- from ~2^17 on the second planes of typical elements run out of fp16 exponent and the error grows: 1.0e-6 at 30000 x); same
    log-sum-exp."""
    from bbdm_amd import _lib
    import kernel_ops as ops
    g = torch.Generator().manual_seed(T + ch)
    C = heads * ch
    qkv = torch.randn(N, T, 3 * C, generator=g) * 1.5
    qkv[0, 3, 5] = 11.0                                         # (an outlier: the bound sits on it)
    x = qkv.double()
    if new_order:
        q, k, v = (z.reshape(N, T, heads, ch) for z in x.chunk(3, dim=2))
    else:
        q, k, v = x.reshape(N, T, heads, 3 * ch).split(ch, dim=3)
    sc = ch ** -0.25
    sim = torch.einsum("nthc,nshc->nhts", q * sc, k * sc)
    ref = torch.einsum("nhts,nshc->nthc", sim.softmax(-1), v).reshape(N, T, C)
    ref_lse = torch.logsumexp(sim, dim=-1)
    qd = qkv.to(dev)
    with _lib.option("attn_pipe", 3):
        a = ops.attention_planes(qd, heads, new_order)
        b, lb = ops.attention_planes(qd, heads, new_order, return_lse=True, bound=float(qkv.abs().max()) * slack)
        lib = _lib.load()
        assert lib.bbdm_attention_kv_planes_h2_bytes(N, T + 32, heads, ch) == 0 and lib.bbdm_attention_kv_planes_h2_bytes(N, T, heads, 16) == 0
        assert 3 * lib.bbdm_attention_kv_planes_h2_bytes(N, T, heads, ch) == 2 * lib.bbdm_attention_kv_planes_bytes(N, T, heads, ch)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    ea, eb = rel_err(a.cpu(), ref), rel_err(b.cpu(), ref)
    print(f"attention N{N} T{T} {heads}x{ch} bound x{slack:g}: bf16x3 {ea:.2e}, fp16 pair {eb:.2e}")
    assert bool(torch.isfinite(b).all()) and eb < TOL and eb < 1.5 * ea + 2e-7, (ea, eb)
    assert rel_err(lb.cpu(), ref_lse) < 1e-5


def test_h2_projection_bound(dev):
    """bbdm_h2_rowl1_f32 + bbdm_h2_affine_bound_f32: bound(W x + b) = bound(x) max_row sum |W| + max |b| -- never below the real maximum of
    the projection, and close to it for the adversarial input (x = bound sign(W[row]))."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(5)
    for rows, cols, has_bias in ((96, 64, True), (3072, 1024, True), (7, 2048, False)):
        w = (torch.randn(rows, cols, 1, 1, generator=g) * 0.1).to(dev)
        b = (torch.randn(rows, generator=g) * 0.5).to(dev) if has_bias else None
        g2 = ops.h2_rowl1(w, b)
        l1 = w.double().abs().sum(dim=(1, 2, 3)).max().item()
        assert l1 <= g2[0].item() <= l1 * 1.01, (l1, g2)
        assert g2[1].item() == (b.abs().max().item() if has_bias else 0.0)
        xb = torch.tensor([3.25], dtype=torch.float32, device=dev)
        ob = ops.h2_affine_bound(xb, g2).item()
        row = int(w.double().abs().sum(dim=(1, 2, 3)).argmax())
        x = 3.25 * torch.sign(w[row].flatten()).double()
        y = (w.double().reshape(rows, cols) @ x.to(w.device)).cpu() + (b.double().cpu() if has_bias else 0.0)
        assert y.abs().max().item() <= ob <= 3.25 * l1 * 1.01 + (g2[1].item() if has_bias else 0.0) + 1e-6


@pytest.mark.parametrize("N", [1, 4, 16, 33, 70])
def test_embedding_path(dev, N):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N)
    dim = 128
    t = torch.randint(0, 1000, (N,), generator=g)
    ref = O.timestep_embedding(t, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    emb = ops.timestep_embedding(t.to(dev), freqs.to(dev), dim)
    torch.cuda.synchronize()
    assert float((emb.cpu() - ref).abs().max()) < 2e-6
    w = torch.randn(520, dim, generator=g) * 0.05
    b = torch.randn(520, generator=g) * 0.1
    y = ops.linear(emb, w.to(dev), b.to(dev), act_in=True, act_out=True)
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), F.silu(F.linear(F.silu(ref), w, b))) < TOL
    w2 = torch.randn(37, 33, generator=g)
    x2 = torch.randn(N, 33, generator=g)
    y2 = ops.linear(x2.to(dev), w2.to(dev), None)
    assert rel_err(y2.cpu(), F.linear(x2, w2)) < TOL


@pytest.mark.parametrize("N,In,Out,act_in,act_out", [(5, 512, 2090, True, False), (33, 512, 2100, True, True), (64, 64, 2050, False, True),
                                                      (16, 128, 40, False, False), (32, 512, 4096, True, False), (32, 512, 512, True, False)])
def test_linear_on_the_matrix_core(dev, N, In, Out, act_in, act_out):
    """bbdm_linear_f32 with In % 8 == 0 and >= 2048 outputs (linear_mfma_kernel): one and two 32-row blocks, ragged 32-output tiles, a K
    that is not a multiple of the four-float4 prefetch group, both activations; below 2048 outputs the thread-per-output kernel."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N + In + Out)
    x = torch.randn(N, In, generator=g)
    w = torch.randn(Out, In, generator=g) * 0.05
    b = torch.randn(Out, generator=g) * 0.1
    ref = F.linear(F.silu(x) if act_in else x, w, b)
    if act_out:
        ref = F.silu(ref)
    y = ops.linear(x.to(dev), w.to(dev), b.to(dev), act_in=act_in, act_out=act_out)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref) < TOL


@pytest.mark.parametrize("N,In,Out,act_in,act_out", [(32, 512, 25088, True, False), (4, 512, 1000, True, False), (17, 64, 40, False, True),
                                                     (32, 96, 33, True, True), (1, 32, 32, False, False)])
def test_linear_on_packed_weights_bitwise(dev, N, In, Out, act_in, act_out):
    """bbdm_linear_packed_f32 (weights packed once into 4 KB blocks that a wave streams sequentially by LDS-DMA, four deep, into LDS of
    its own) against F.linear and, bit for bit, against bbdm_linear_f32 -- the FiLM projection of the reference UNets (512 -> 25088),
    ragged output tiles, one K block and many, a padding wave in the last workgroup, both activations."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(N + In + Out)
    x = torch.randn(N, In, generator=g)
    w = torch.randn(Out, In, generator=g) * 0.05
    b = torch.randn(Out, generator=g) * 0.1
    ref = F.linear(F.silu(x) if act_in else x, w, b)
    if act_out:
        ref = F.silu(ref)
    y = ops.linear_packed(x.to(dev), w.to(dev), b.to(dev), act_in=act_in, act_out=act_out).cpu()
    y0 = ops.linear(x.to(dev), w.to(dev), b.to(dev), act_in=act_in, act_out=act_out).cpu()
    assert rel_err(y, ref) < TOL
    if In % 8 == 0 and In >= 64 and Out >= 2048:        # bbdm_linear_f32 takes its matrix-core kernel: the same MFMA steps in the same order
        assert torch.equal(y, y0), (y - y0).abs().max()
    else:
        assert rel_err(y, y0) < TOL


@pytest.mark.parametrize("objective", ["grad", "noise", "ysubx"])
def test_bridge_arithmetic(dev, objective):
    """q_sample / predict_x0 / p_sample update / loss against the oracle formulas: bit-exact or 1 ulp."""
    import argparse
    import bbdm_amd
    g = torch.Generator().manual_seed(21)
    N, C, S = 3, 3, 16
    up = dict(image_size=S, in_channels=6, out_channels=3, model_channels=32, channel_mult=(1,), num_res_blocks=1,
              attention_resolutions=(), num_head_channels=32, use_scale_shift_norm=True, resblock_updown=True,
              condition_key="SpatialRescaler")
    bb = dict(num_timesteps=1000, mt_type="linear", max_var=1.0, eta=0.8, skip_sample=True, sample_type="linear",
              sample_step=200, loss_type="l1", objective=objective, UNetParams=argparse.Namespace(**up))
    m = bbdm_amd.BrownianBridgeModel(argparse.Namespace(BB=argparse.Namespace(params=argparse.Namespace(**bb)))).to(dev)
    bufs, steps = O.make_schedule(1000, "linear", 1.0, True, "linear", 200)
    x0 = torch.randn(N, C, S, S, generator=g)
    y = torch.randn(N, C, S, S, generator=g)
    eps = torch.randn(N, C, S, S, generator=g)
    pred = torch.randn(N, C, S, S, generator=g)
    t = torch.tensor([0, 517, 999])
    x_t, tgt = m.q_sample(x0.to(dev), y.to(dev), t.to(dev), eps.to(dev))
    x_t_ref, tgt_ref = O.q_sample(bufs, x0, y, t, eps, objective)
    assert rel_err(x_t.cpu(), x_t_ref) < 1e-6 and rel_err(tgt.cpu(), tgt_ref) < 1e-6
    x0r = m.predict_x0_from_objective(x_t_ref.to(dev), y.to(dev), t.to(dev), pred.to(dev))
    assert rel_err(x0r.cpu(), O.predict_x0(bufs, x_t_ref, y, t, pred, objective)) < 1e-6
    from bbdm_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    for i in (0, 1, 100, 198, 199):
        for clip in (0, 1):
            a_ref, b_ref = O.p_sample_update(bufs, steps, i, x_t_ref, y, pred, eps, objective, 0.8, bool(clip))
            xg, yg, pg, eg = (z.to(dev) for z in (x_t_ref, y, pred, eps))
            a, b = torch.empty_like(xg), torch.empty_like(xg)
            alias = torch.full_like(xg, float("nan")) if clip else None      # the second x_next destination (the next step's input buffer)
            last = int(steps[i]) == 0
            _lib.call("bbdm_bb_p_sample_step_f32", xg.data_ptr(), yg.data_ptr(), pg.data_ptr(), eg.data_ptr(),
                      m.m_t.data_ptr(), m.variance_t.data_ptr(), int(steps[i]), 0 if last else int(steps[i + 1]),
                      int(last), 0.8, clip, {"grad": 0, "noise": 1, "ysubx": 2}[objective], a.data_ptr(),
                      b.data_ptr(), None if alias is None else alias.data_ptr(), N, C * S * S, st)
            torch.cuda.synchronize()
            assert rel_err(a.cpu(), a_ref) < 2e-6 and rel_err(b.cpu(), b_ref) < 2e-6, (i, clip)
            assert alias is None or torch.equal(alias, a)
    for lt in ("l1", "l2"):
        m.loss_type = lt
        got = float(m._loss(tgt_ref.to(dev), pred.to(dev)))
        want = float(O.bb_loss(tgt_ref, pred, lt))
        assert abs(got - want) < 1e-6 * max(1.0, abs(want))


def test_layout_roundtrip(dev):
    import kernel_ops as ops
    a, b = torch.randn(2, 3, 5, 7), torch.randn(2, 3, 5, 7)
    x = ops.nchw_to_nhwc(a.to(dev), b.to(dev))
    assert x.shape == (2, 5, 7, 8)
    ref = torch.cat([a, b, torch.zeros(2, 2, 5, 7)], 1)
    assert torch.equal(x.cpu().permute(0, 3, 1, 2), ref)
    back = ops.nhwc_to_nchw(x, 6)
    assert torch.equal(back.cpu(), ref[:, :6])


@pytest.mark.parametrize("N,H,W,C,Cout,ks,film,silu", [(2, 16, 16, 128, 64, 3, True, True), (3, 8, 8, 96, 128, 3, False, True),
                                                        (1, 32, 32, 256, 128, 1, False, False), (20, 4, 4, 640, 256, 3, True, True),
                                                        (1, 64, 64, 64, 3, 3, False, True),        # narrow head + fused GN
                                                        (2, 72, 88, 32, 3, 3, True, True)])        # ... ragged 16x16 tiles, two chunks, FiLM
def test_conv_with_fused_groupnorm_producer(dev, N, H, W, C, Cout, ks, film, silu):
    """GroupNorm -> [FiLM] -> [SiLU] -> conv with the normalisation applied while the patch is staged."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(C + Cout + ks)
    x = torch.randn(N, C, H, W, generator=g) * 1.7 + 0.4
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    fl = 0.3 * torch.randn(N, 2 * C, generator=g) if film else None
    w = torch.randn(Cout, C, ks, ks, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    y = F.group_norm(x, 32, gamma, beta, 1e-5)
    if film:
        y = y * (1 + fl[:, :C, None, None]) + fl[:, C:, None, None]
    if silu:
        y = F.silu(y)
    ref = F.conv2d(y, w, b, padding=ks // 2)
    xg = _nhwc(x).to(dev)
    stats = ops.groupnorm_stats(xg)
    sc, bi = ops.groupnorm_coeffs(stats, gamma.to(dev), beta.to(dev), H * W, film=fl.to(dev) if film else None)
    out = ops.conv2d_nhwc(xg, ops.pack_conv_weight(w.to(dev)), b.to(dev), Cout, ks, pre_scale=sc, pre_bias=bi,
                          pre_silu=silu)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out.cpu()), ref) < TOL


@pytest.mark.parametrize("m,Cout,Cin,in_pad,dgrad", [(4, 96, 40, 48, False), (6, 128, 128, 128, False), (2, 24, 16, 32, True),
                                                      (4, 200, 72, 208, True), (6, 40, 130, 144, False)])
def test_winograd_weight_planes_fused_bitwise(dev, m, Cout, Cin, in_pad, dgrad):
    """bbdm_winograd_pack_weight_bf3p_f32 (G g G^T straight into the bf16 planes) against the two launches it replaces: ragged
    Cout / Cin (zero rows and k), the data-gradient orientation, padded input channels.  Same layout, same zeros; the fp32 value each
    plane triple adds up to is the same up to the compiler's FMA contraction of G g G^T in the two kernel bodies (<= 1 ulp of the
    largest term on the GPU; bit-identical on the emulator)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(m + Cout + Cin)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.3).to(dev)
    a = ops.winograd_weight_planes(w, m, in_pad, dgrad, fused=True)
    b = ops.winograd_weight_planes(w, m, in_pad, dgrad, fused=False)
    if dev.type == "cuda":
        torch.cuda.synchronize()

    def value(planes):                  # [units][3 planes][512 bf16] -> the fp32 numbers the three planes add up to
        h = planes.cpu().view(torch.int16).view(-1, 3, 512).view(torch.bfloat16).float()
        return (h[:, 0] + h[:, 1]) + h[:, 2]

    va, vb = value(a), value(b)
    assert torch.equal(va == 0, vb == 0)
    assert float((va - vb).abs().max()) <= 2.0 ** -22 * float(vb.abs().max())
    if dev.type != "cuda":
        assert torch.equal(a.cpu(), b.cpu())
