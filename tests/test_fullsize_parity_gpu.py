"""Parity at the REAL BASELINE.json configurations (GPU; round-1 VERDICT "weak" #1: every model-level parity test ran at
<= 64x64 while the headline number is quoted at 256x256).

 * C2  pixel BBDM 256x256: the 237 M-parameter Template-BBDM UNet, one ``p_sample`` step at 256x256 against the CPU
       oracle, with the direct kernel only (winograd = 0) and with the default plan (F(4x4,3x3)), batch 2 and batch 1.
       This is the first test that runs T = 4096 attention, the 8x32 spatial tile at W = 256 and the million-pixel
       index ranges inside the model.
 * attention kernel alone at T = 4096, 16 heads x 64 channels, against an fp64 einsum/softmax.
 * C4  LBBDM-f4 training: loss + all 248 parameter gradients of the full-size UNet (latent 3x64x64, batch 2) against
       torch.autograd on the oracle.
 * C5  LBBDM-f16: the real f16 template (in/out 8 channels, attention at ds = 1, 2, 4 -> 6 attention blocks, 258 M
       parameters), one step at 16x16, batch 2.

Bar: 1e-3 relative per sampling step (BASELINE.json north_star); the measured errors are printed and quoted in
DESIGN.md §5.  The oracle needs ~10 s per 256x256 image on the GPU box's host cores."""
import argparse
import math

import pytest
import torch

import bbdm_oracle as O
from fixture_weights import synth_weights
from fixtures import rel_err

pytestmark = pytest.mark.gpu

UNET_PIXEL = dict(in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(32, 16, 8),
                  channel_mult=(1, 4, 8), conv_resample=True, dims=2, num_heads=8, num_head_channels=64,
                  use_scale_shift_norm=True, resblock_updown=True, use_spatial_transformer=False, context_dim=None,
                  condition_key="SpatialRescaler")
BB = dict(mt_type="linear", objective="grad", loss_type="l1", skip_sample=True, sample_type="linear",
          sample_step=200, num_timesteps=1000, eta=1.0, max_var=1.0)


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(up, bb, seed, dev):
    import bbdm_amd
    m = bbdm_amd.BrownianBridgeModel(_ns({"BB": {"params": dict(bb, UNetParams=up)}}))
    shapes = [(k, tuple(v.shape)) for k, v in m.denoise_fn.state_dict().items()]
    sd = synth_weights(shapes, seed, w_std=0.02)
    m.denoise_fn.load_state_dict(sd, strict=True)
    return m.to(dev), sd


def _p_sample(m, x_t, y, ctx, i, eps, dev):
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: eps.to(dev)
    try:
        a, b = m.p_sample(x_t.to(dev), y.to(dev), None if ctx is None else ctx.to(dev), i, clip_denoised=False)
    finally:
        torch.randn_like = orig
    torch.cuda.synchronize()
    return a.cpu(), b.cpu()


def test_c2_256x256_step_direct_and_winograd(dev):
    """BASELINE.json configs[1] at its real resolution (BrownianBridgeModel.py:171-201 over openaimodel.py:721-759)."""
    up = dict(UNET_PIXEL, image_size=256)
    m, sd = _model(up, dict(BB, skip_sample=False), 777, dev)
    m.eval()
    assert sum(v.numel() for v in sd.values()) == 237094787
    g = torch.Generator().manual_seed(1234)
    N, S = 2, 256
    y = torch.randn(N, 3, S, S, generator=g).clamp(-1, 1)
    x_t = torch.randn(N, 3, S, S, generator=g).clamp(-1, 1)
    eps = torch.randn(N, 3, S, S, generator=g)
    ora = O.OracleBBDM({"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**up), **dict(BB, skip_sample=False))
    i = 431
    with torch.no_grad():
        a_ref, b_ref = ora.p_sample(x_t, y, y, i, clip_denoised=False, noise=eps)
    for wino in (0, 4, 6):
        m.denoise_fn.winograd = wino
        m.denoise_fn._plans = {}                          # one 256^2 plan resident at a time
        for n in (N, 1):
            a, b = _p_sample(m, x_t[:n], y[:n], y[:n], i, eps[:n], dev)
            plan = next(iter(m.denoise_fn._plans.values())) if n == N else None
            if plan is not None:
                n_wino = sum(name == "bbdm_winograd_gemm_f32" for name, _ in plan.ops)
                assert (n_wino == 0) == (wino == 0)
                t_attn = [args[6] for name, args in plan.ops if name == "bbdm_attention_f32"]
                assert t_attn == [4096]
            ea, eb = rel_err(a, a_ref[:n]), rel_err(b, b_ref[:n])
            print(f"C2 256x256 step, batch {n}, winograd={wino}: rel err x_tminus {ea:.2e}  x0_recon {eb:.2e}")
            assert ea < 1e-3 and eb < 1e-3
            m.denoise_fn._plans = {}
    torch.cuda.empty_cache()


@pytest.mark.parametrize("new_order", [False, True])
def test_attention_T4096(dev, new_order):
    """QKVAttentionLegacy / QKVAttention (openaimodel.py:359-375, 398-413) at the C2 middle-block shape: T = 64*64."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(4096)
    N, T, heads, ch = 1, 4096, 16, 64
    C = heads * ch
    qkv = torch.randn(N, 3 * C, T, generator=g) * 1.5
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(N * heads, ch, T) for z in (q, k, v))
    else:
        q, k, v = qkv.reshape(N * heads, 3 * ch, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    ref = torch.empty(N * heads, ch, T, dtype=torch.float64)
    for h in range(N * heads):                            # one head at a time: 134 MB of fp64 scores
        w = torch.softmax(torch.einsum("ct,cs->ts", (q[h] * s).double(), (k[h] * s).double()), dim=-1)
        ref[h] = torch.einsum("ts,cs->ct", w, v[h].double())
    ref = ref.reshape(N, C, T)
    out = ops.attention(qkv.permute(0, 2, 1).contiguous().to(dev), heads, new_order)
    torch.cuda.synchronize()
    e = rel_err(out.cpu().permute(0, 2, 1), ref)
    print(f"attention T=4096 heads=16 ch=64 new_order={new_order}: rel err {e:.2e}")
    assert e < 1e-5


def _oracle_loss_grads(sd, up, bb, x0, y, t, nz, dtype):
    sd_o = {"denoise_fn." + k: v.to(dtype).clone().requires_grad_() for k, v in sd.items()}
    ora = O.OracleBBDM(sd_o, O.UNetSpec(**up), **bb)
    lo, _ = ora.p_losses(x0.to(dtype), y.to(dtype), None, t, nz.to(dtype))
    lo.backward()
    return float(lo.detach()), {k[len("denoise_fn."):]: v.grad for k, v in sd_o.items()}


def test_c4_full_size_loss_and_all_gradients(dev):
    """BASELINE.json configs[3] per-GPU work at the real model size: LBBDM-f4 UNet (in 3, nocond), latent 3x64x64, batch 2 --
    loss and every one of the 248 parameter gradients against autograd on the oracle (BrownianBridgeModel.py:98-126).

    At this size a handful of gradients (GroupNorm gains of wide layers, FiLM projections: sums of thousands of
    cancelling terms) are conditioned worse than 1e-3 in fp32 -- the oracle's OWN fp32 gradients differ from an fp64
    evaluation of the same graph by more than that.  So the oracle is evaluated twice (fp32 = the reference's arithmetic,
    fp64 = the exact value) and a parameter passes when the HIP gradient is within 1e-3 of the fp32 oracle, OR is as close
    to the fp64 value as the fp32 oracle itself is (factor 4).  Errors are scaled by max(|g|_max, 1e-3 * largest |g| in
    the model), as in tests/test_training_gpu.py."""
    up = dict(UNET_PIXEL, image_size=64, in_channels=3, condition_key="nocond")
    m, sd = _model(up, BB, 4040, dev)
    m.train()
    g = torch.Generator().manual_seed(77)
    N = 2
    x0 = torch.randn(N, 3, 64, 64, generator=g)
    y = torch.randn(N, 3, 64, 64, generator=g)
    t = torch.tensor([812, 37])
    nz = torch.randn(N, 3, 64, 64, generator=g)
    l32, g32 = _oracle_loss_grads(sd, up, BB, x0, y, t, nz, torch.float32)
    l64, g64 = _oracle_loss_grads(sd, up, BB, x0, y, t, nz, torch.float64)
    assert len(g32) == 248
    gmax = max(float(v.abs().max()) for v in g64.values())
    scale = {k: max(float(v.abs().max()), 1e-3 * gmax) for k, v in g64.items()}
    e_ora = {k: float((g32[k].double() - g64[k]).abs().max()) / scale[k] for k in g64}
    worst_ora = sorted(((e, k) for k, e in e_ora.items()), reverse=True)[:3]
    print(f"C4 full-size: oracle fp32 vs fp64 loss {l32:.7f} / {l64:.7f}; worst fp32-oracle gradient errors vs fp64: " +
          "; ".join(f"{k} {e:.2e}" for e, k in worst_ora))
    failures = []
    for wino in (6, 4, 0):
        m.denoise_fn.winograd = wino
        m.denoise_fn._plans = {}
        m.zero_grad(set_to_none=True)
        loss, _ = m.p_losses(x0.to(dev), y.to(dev), None, t.to(dev), nz.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        lv = float(loss.detach())
        assert abs(lv - l32) < 1e-5 * max(1.0, abs(l32))
        rows = []
        for k, p in m.denoise_fn.named_parameters():
            assert p.grad is not None, k
            gg = p.grad.cpu()
            e32 = float((gg - g32[k]).abs().max()) / scale[k]
            e64 = float((gg.double() - g64[k]).abs().max()) / scale[k]
            ok = e32 < 1e-3 or e64 <= 4.0 * e_ora[k] + 1e-6
            rows.append((e32, e64, e_ora[k], k, ok))
        rows.sort(reverse=True)
        n_tol = sum(r[0] < 1e-3 for r in rows)
        print(f"C4 full-size gradients (237 M, batch {N}), winograd={wino}: loss {lv:.6f}; {n_tol}/248 within 1e-3 of the "
              "fp32 oracle; worst (vs fp32 oracle | vs fp64 | fp32 oracle vs fp64): " +
              "; ".join(f"{k} {a:.2e}|{b:.2e}|{c:.2e}" for a, b, c, k, _ in rows[:4]))
        failures += [(wino, k, a, b, c) for a, b, c, k, ok in rows if not ok]
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()
    assert not failures, failures[:5]


def test_c5_real_f16_template_step(dev):
    """BASELINE.json configs[4]: Template-LBBDM-f16.yaml's UNet exactly (configs/Template-LBBDM-f16.yaml:107-130)."""
    up = dict(UNET_PIXEL, image_size=16, in_channels=8, out_channels=8, attention_resolutions=(16, 8, 4),
              condition_key="nocond")
    m, sd = _model(up, BB, 1616, dev)
    m.eval()
    n_attn = sum(".qkv.weight" in k for k in sd)
    nparam = sum(v.numel() for v in sd.values())
    assert n_attn == 6, n_attn
    g = torch.Generator().manual_seed(16)
    N = 2
    y = torch.randn(N, 8, 16, 16, generator=g)
    x_t = torch.randn(N, 8, 16, 16, generator=g)
    eps = torch.randn(N, 8, 16, 16, generator=g)
    ora = O.OracleBBDM({"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**up), **BB)
    for i in (0, 100, 199):
        with torch.no_grad():
            a_ref, b_ref = ora.p_sample(x_t, y, None, i, clip_denoised=False, noise=eps)
        a, b = _p_sample(m, x_t, y, None, i, eps, dev)
        ea, eb = rel_err(a, a_ref), rel_err(b, b_ref)
        print(f"C5 f16 template ({nparam / 1e6:.1f} M params, 6 attention blocks) step i={i}: rel err {ea:.2e} {eb:.2e}")
        assert ea < 1e-3 and eb < 1e-3
