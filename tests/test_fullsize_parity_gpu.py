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
from fixtures import parity_err, rel_err

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
    for wino in (0, 4, 6, 8):                             # (8, the default: F(8x8, 3x3) on the large layers, F(6x6) / F(7x7, 2x2) elsewhere)
        m.denoise_fn.winograd = wino
        m.denoise_fn._plans = {}                          # one 256^2 plan resident at a time
        for n in (N, 1):
            a, b = _p_sample(m, x_t[:n], y[:n], y[:n], i, eps[:n], dev)
            plan = next(iter(m.denoise_fn._plans.values())) if n == N else None
            if plan is not None:
                n_wino = sum(name == "bbdm_winograd_gemm_f32" for name, _ in plan.ops)
                assert (n_wino == 0) == (wino == 0)
                tiles = {args[0] for name, args in plan.ops if name == "bbdm_winograd_gemm_f32"}
                assert (8 in tiles) == (wino == 8) and (not tiles or max(tiles) <= max(wino, 7 if wino >= 6 else 0)), tiles
                t_attn = [args[6] for name, args in plan.ops if name == "bbdm_attention_f32"]
                assert t_attn == [4096]
            ea, eb = parity_err(a, a_ref[:n]), parity_err(b, b_ref[:n])
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
    e = parity_err(out.cpu().permute(0, 2, 1), ref)
    print(f"attention T=4096 heads=16 ch=64 new_order={new_order}: rel err {e:.2e}")
    assert e < 1e-5


def _oracle_grads(sd, up, bb, x0, y, t, nz, sign=None):
    """Autograd on the oracle.  ``sign`` = None: the model's own loss (BrownianBridgeModel.py:98-126).  ``sign`` given
    (l1 only): d|target - pred| / d pred is taken with THAT sign pattern, i.e. loss = sum(sign * (target - pred)) / count --
    the L1 loss with the non-differentiable sign() frozen (see the test's docstring)."""
    sd_o = {"denoise_fn." + k: v.clone().requires_grad_() for k, v in sd.items()}
    ora = O.OracleBBDM(sd_o, O.UNetSpec(**up), **bb)
    x_t, target = O.q_sample(ora.bufs, x0, y, t, nz, ora.objective)
    pred = ora.denoise(x_t, t, None)
    if sign is None:
        lo = O.bb_loss(target, pred, ora.loss_type)
    else:
        lo = (sign * (target - pred)).sum() / pred.numel()
    lo.backward()
    return float(lo.detach()), pred.detach(), target, {k[len("denoise_fn."):]: v.grad for k, v in sd_o.items()}


@pytest.mark.parametrize("loss_type", ["l2", "l1"])
def test_c4_full_size_loss_and_all_gradients(dev, loss_type):
    """BASELINE.json configs[3] per-GPU work at the real model size: LBBDM-f4 UNet (in 3, nocond), latent 3x64x64, batch 2 --
    loss and every one of the 248 parameter gradients against autograd on the oracle (BrownianBridgeModel.py:98-126), with
    the Winograd plan (default), F(4x4) and the direct kernels.

    'l2' is compared as is.  For 'l1' (the templates' loss) d loss / d pred = -sign(target - pred) / count is discontinuous:
    of the 24 576 output elements a few have |target - pred| below the fp32 forward difference between the two
    implementations (~1e-5), their sign flips, and ONE flipped element already moves every gradient of the network by
    ~1e-3 of its magnitude (round-2 diagnosis: gradients were 2e-3..2e-2 off in every layer, cosine 0.99999, while the fp32
    oracle is within 3e-6 of an fp64 evaluation and every backward kernel passes at these shapes).  That is a property of
    the loss, not of the backward pass, so the oracle is differentiated with the HIP forward's sign pattern (the number of
    flipped elements is printed and asserted <= 8); the loss VALUE is compared unmodified."""
    up = dict(UNET_PIXEL, image_size=64, in_channels=3, condition_key="nocond")
    bb = dict(BB, loss_type=loss_type)
    m, sd = _model(up, bb, 4040, dev)
    m.train()
    g = torch.Generator().manual_seed(77)
    N = 2
    x0 = torch.randn(N, 3, 64, 64, generator=g)
    y = torch.randn(N, 3, 64, 64, generator=g)
    t = torch.tensor([812, 37])
    nz = torch.randn(N, 3, 64, 64, generator=g)
    l_ref, pred_ref, target, g_plain = _oracle_grads(sd, up, bb, x0, y, t, nz)
    worst_all = 0.0
    for wino in (6, 4, 0):
        m.denoise_fn.winograd = wino
        m.denoise_fn._plans = {}
        m.zero_grad(set_to_none=True)
        seen = {}
        hook = m.denoise_fn.register_forward_hook(lambda mod, inp, out: seen.__setitem__("pred", out.detach()))
        loss, log = m.p_losses(x0.to(dev), y.to(dev), None, t.to(dev), nz.to(dev))
        hook.remove()
        loss.backward()
        torch.cuda.synchronize()
        lv = float(loss.detach())
        assert abs(lv - l_ref) < 1e-5 * max(1.0, abs(l_ref))
        g_ref, flips = g_plain, 0
        if loss_type == "l1":
            # the sign pattern the HIP loss kernel differentiated with: its OWN target and prediction (the UNet output, caught by a
            # forward hook), whose fp32 difference has an exact sign.  (Recovering the prediction from the logged x0_recon = x_t - pred
            # is one rounding away from it: an element with |target - pred| ~ 1e-7 then gets the other sign, and ONE such element is a
            # 1e-2 disagreement in some gradients -- seen once the embedding path moved to the matrix core and the forward by 1e-7.)
            with torch.no_grad():
                _, target_gpu = m.q_sample(x0.to(dev), y.to(dev), t.to(dev), nz.to(dev))
            pred_gpu = seen["pred"].float().cpu()
            sign = torch.sign(target_gpu.cpu() - pred_gpu)
            flips = int((sign != torch.sign(target - pred_ref)).sum())
            # ... and BOUNDED: a forward regression must not hide behind the frozen sign pattern.  With |target - pred| ~ 1 and a
            # forward difference of ~1e-5, 24 576 elements give 0-2 flips (measured); 8 already means the forward moved by ~1e-4.
            assert flips <= 8, flips
            _, _, _, g_ref = _oracle_grads(sd, up, bb, x0, y, t, nz, sign=sign)
        gmax = max(float(v.abs().max()) for v in g_ref.values())
        rows = []
        for k, p in m.denoise_fn.named_parameters():
            assert p.grad is not None, k
            ref = g_ref[k]
            rows.append((float((p.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * gmax), k))
        rows.sort(reverse=True)
        worst_all = max(worst_all, rows[0][0])
        print(f"C4 full-size gradients (237 M, batch {N}, {loss_type}), winograd={wino}: loss {lv:.6f} (oracle {l_ref:.6f}); "
              f"sign flips {flips}/{target.numel()}; worst of 248: " + "; ".join(f"{k} {e:.2e}" for e, k in rows[:3]))
        assert rows[0][0] < 1e-3, rows[0]
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()


@pytest.mark.parametrize("workload", ["c2", "c3", "c5", "c1"])
def test_benchmarked_plan_exactly(dev, workload):
    """The plans bench.py times, not a smaller batch of the same model: C2 = the headline, pixel 256x256, batch 16 (bench.py's own
    `parity` field compares image 0 of the timed batch; here images 0 AND 15); C3 = LBBDM-f4 UNet, latent 3x64x64, batch 32, nocond;
    C5 = the f16 UNet, 8x16x16, batch 32; C1 = pixel 64x64, batch 4 -- each replayed as a hipGraph as in the benchmark (tile choices,
    split-K and the graph path depend on the batch).  One p_sample step through the graph-replayed plan, images 0 and N - 1 against
    the oracle."""
    import bench
    desc, up, ch, size, batch, skip, sstep = bench.WORKLOADS[workload]
    bb = dict(BB, skip_sample=skip, sample_step=sstep)
    m, sd = _model(up, bb, 3131, dev)
    m.eval()
    g = torch.Generator().manual_seed(31 + batch)
    y = torch.randn(batch, ch, size, size, generator=g).clamp(-1, 1)
    x_t = torch.randn(batch, ch, size, size, generator=g).clamp(-1, 1)
    eps = torch.randn(batch, ch, size, size, generator=g)
    ctx = None if up["condition_key"] == "nocond" else y
    ora = O.OracleBBDM({"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**up), **bb)
    i = 57
    for rep in range(2):                                  # the second call replays the captured graph
        a, b = _p_sample(m, x_t, y, ctx, i, eps, dev)
    plan = next(iter(m.denoise_fn._plans.values()))
    assert plan.N == batch and plan._want_graph() and plan._graph is not None, (plan.N, plan._want_graph())
    for row in (0, batch - 1):
        sl = slice(row, row + 1)
        with torch.no_grad():
            a_ref, b_ref = ora.p_sample(x_t[sl], y[sl], None if ctx is None else y[sl], i, clip_denoised=False, noise=eps[sl])
        ea, eb = parity_err(a[sl], a_ref), parity_err(b[sl], b_ref)
        print(f"{workload} at the benchmarked plan (batch {batch}, hipGraph), image {row}: rel err {ea:.2e} {eb:.2e}")
        assert ea < 1e-3 and eb < 1e-3
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()


@pytest.mark.parametrize("loss_type", ["l2", "l1"])
def test_c4_benchmarked_training_plan_batch32(dev, loss_type):
    """(``loss_type`` "l1", the templates' loss, round 6: with the frozen sign pattern of test_c4_full_size_loss_and_all_gradients --
    the F(8x8, 3x3) gradient side had no full-size l1 check.)
    BASELINE.json configs[3] at the plan bench.py times: LBBDM-f4 training, latent 3x64x64, batch 32 per GPU (the full-size gradient
    test above runs batch 2: batch 32 selects other GEMM tiles, split-K counts and Winograd weight-gradient shapes).  Loss and eight
    ALL 248 parameter gradients (bench.py's line carries eight named ones: stem, a 512-channel 3x3 layer, qkv, a middle-block out conv,
    a 1x1 skip connection, the embedding MLP, the head) of ONE micro-step against autograd on the oracle over the whole batch (l2 loss: see the docstring above for why
    the l1 gradients need a frozen sign pattern); bench.py's `parity` of the c4 line is this same comparison."""
    import bench
    desc, up, ch, size, batch, skip, sstep = bench.WORKLOADS["c4"]
    bb = dict(BB, skip_sample=skip, sample_step=sstep)
    m, sd = _model(up, bb, 3232, dev)
    m.train()
    x0, y = bench.make_inputs(batch, ch, size, seed=99)
    par = bench.training_parity(m, sd, up, skip, sstep, x0.to(dev), y.to(dev), dev, all_grads=True, loss_type=loss_type)
    assert loss_type == "l2" or 0 <= par["sign_flips"] <= 64, par["sign_flips"]      # (393 216 loss terms; measured: a handful)
    plan = next(iter(m.denoise_fn._plans.values()))
    assert plan.N == batch == 32 and plan.training and len(par["grad_errors"]) == 248
    # the benchmarked plan runs its 64^2 and 32^2 levels on F(8x8, 3x3) in all three directions (UNetModel.winograd_train8; batch 2 of
    # the full-size test above has too few tiles for it)
    for ops_, lo in ((plan.ops, 20), (plan.bops, 20)):
        assert sum(1 for n, a in ops_ if str(n) == "bbdm_winograd_gemm_f32" and a[0] == 8) >= lo
    assert sum(1 for n, a in plan.bops if str(n) == "bbdm_winograd_wgrad_finish_bias_f32" and a[0] == 8) >= 20
    print(f"c4 at the benchmarked training plan (batch {batch}, {loss_type}, sign flips {par['sign_flips']}): loss rel err "
          f"{par['rel_err_loss']:.2e}, worst of all 248 gradients {par['rel_err_grad_worst']:.2e} ({par['cpu_seconds']:.0f} s of oracle autograd)")
    assert par["rel_err_loss"] < 1e-5, par
    assert par["rel_err_grad_worst"] < 1e-3, par["grad_errors"]
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()


@pytest.mark.parametrize("workload", ["c2", "c3", "c5"])
def test_step_is_bitwise_reproducible(dev, workload):
    """The reference seeds everything and sets cudnn.deterministic (main.py:57-65): a sampling step must give the SAME BITS for the same
    inputs.  The only order-dependent reductions of the sampling path are the GroupNorm statistics, which many workgroups of the
    producing kernels add up: they are accumulated as integer limbs (csrc/stats_acc.h: integer addition is associative, so the order of
    the atomics cannot change the sum), which makes the step bitwise reproducible by construction.  C2 (the headline plan: T = 4096
    attention, the default F(8x8) / F(7x7,2x2) tiles on the fp16-pair planes, batch 16 at 256x256) / C3 / C5 plans at their benchmarked batch: the eager warm-up call, the call that
    captures the hipGraph and two replays must agree bit for bit."""
    import bench
    desc, up, ch, size, batch, skip, sstep = bench.WORKLOADS[workload]
    bb = dict(BB, skip_sample=skip, sample_step=sstep)
    m, sd = _model(up, bb, 777, dev)
    m.eval()
    g = torch.Generator().manual_seed(5)
    y = torch.randn(batch, ch, size, size, generator=g).clamp(-1, 1)
    x_t = torch.randn(batch, ch, size, size, generator=g).clamp(-1, 1)
    eps = torch.randn(batch, ch, size, size, generator=g)
    ctx = None if up["condition_key"] == "nocond" else y
    outs = [_p_sample(m, x_t, y, ctx, 57, eps, dev) for _ in range(4)]
    for k, (a, b) in enumerate(outs[1:], 1):
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1]), \
            (k, float((a - outs[0][0]).abs().max()), float((b - outs[0][1]).abs().max()))
    # ... and with the launches issued one by one instead of replayed (same kernels, other timing)
    m.denoise_fn.hip_graph = False
    a, b = _p_sample(m, x_t, y, ctx, 57, eps, dev)
    assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()


@pytest.mark.parametrize("loss_type", ["l1", "l2"])
def test_training_step_is_bitwise_reproducible(dev, loss_type):
    """The reference sets cudnn.deterministic for training as well (main.py:57-65).  BASELINE.json configs[3] at the benchmarked plan
    (LBBDM-f4 UNet, latent 3x64x64, batch 32): three micro-steps from the same state -- same seed for the timesteps and the noise --
    must give the same bits in the loss and in EVERY parameter gradient.  The backward's order-dependent sums (GroupNorm dgamma / dbeta
    and the per-group sums, bias gradients, the loss) are accumulated as integer limbs (csrc/stats_acc.h) or in a fixed order; split-K
    partial sums of the weight-gradient GEMMs were added in a fixed order already."""
    import bench
    desc, up, ch, size, batch, skip, sstep = bench.WORKLOADS["c4"]
    bb = dict(BB, skip_sample=skip, sample_step=sstep, loss_type=loss_type)
    m, sd = _model(up, bb, 4242, dev)
    m.train()
    x0, y = bench.make_inputs(batch, ch, size, seed=77)
    x0, y = x0.to(dev), y.to(dev)
    runs = []
    for rep in range(3):
        for p in m.parameters():
            p.grad = None
        torch.manual_seed(1234)
        loss, _ = m(x0, y)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert len(runs[0][1]) == 248
    for rep in (1, 2):
        assert torch.equal(runs[rep][0], runs[0][0]), (rep, float(runs[rep][0]), float(runs[0][0]))
        bad = [k for k, g in runs[rep][1].items() if not torch.equal(g, runs[0][1][k])]
        assert not bad, (rep, len(bad), bad[:8])
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()


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
        ea, eb = parity_err(a, a_ref), parity_err(b, b_ref)
        print(f"C5 f16 template ({nparam / 1e6:.1f} M params, 6 attention blocks) step i={i}: rel err {ea:.2e} {eb:.2e}")
        assert ea < 1e-3 and eb < 1e-3


def test_f8_template_step(dev):
    """configs/Template-LBBDM-f8.yaml:106-129 -- the one reference template without a full-size step test (round-5 verdict, item 9): latent
    4 x 32 x 32, in / out 4 channels, 237 M parameters, 200-step schedule.  One p_sample step at batch 2 and at batch 32 (the plans differ:
    tile choices follow the batch) against the oracle, three schedule indices."""
    up = dict(UNET_PIXEL, image_size=32, in_channels=4, out_channels=4, condition_key="nocond")
    m, sd = _model(up, BB, 808, dev)
    m.eval()
    ora = O.OracleBBDM({"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**up), **BB)
    g = torch.Generator().manual_seed(8)
    for N in (2, 32):
        y = torch.randn(N, 4, 32, 32, generator=g)
        x_t = torch.randn(N, 4, 32, 32, generator=g)
        eps = torch.randn(N, 4, 32, 32, generator=g)
        for i in (0, 100, 199):
            a, b = _p_sample(m, x_t, y, None, i, eps, dev)
            with torch.no_grad():
                a_ref, b_ref = ora.p_sample(x_t[:2], y[:2], None, i, clip_denoised=False, noise=eps[:2])
            ea, eb = parity_err(a[:2], a_ref), parity_err(b[:2], b_ref)
            print(f"LBBDM-f8 template (latent 4x32x32), batch {N}, step i={i}: rel err {ea:.2e} {eb:.2e}")
            assert ea < 1e-3 and eb < 1e-3
    m.denoise_fn._plans = {}
    torch.cuda.empty_cache()
