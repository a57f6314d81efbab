"""Training path (GPU): loss and every parameter gradient of bbdm_amd against torch.autograd on the CPU oracle; a short
Adam run; the reference's DDP wrapping (world size 1 over RCCL) driving the custom autograd node."""
import argparse
import os

import pytest
import torch

import bbdm_oracle as O
from fixtures import load_case, oracle_model, rel_err

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-3          # per-parameter: max|g - g_ref| / max|g_ref|
CASES = ("tiny_concat", "tiny_nocond", "tiny_ysubx", "tiny_xattn")       # tiny_xattn: SpatialTransformer blocks


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


def build(rec, dev):
    import bbdm_amd
    m = bbdm_amd.BrownianBridgeModel(_ns({"BB": {"params": dict(rec["bb_params"], UNetParams=rec["unet_params"])}}))
    m.load_state_dict(rec["state_dict"], strict=True)
    return m.to(dev)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_grads(rec):
    rec = dict(rec)
    rec["state_dict"] = {k: (v.clone().requires_grad_() if k.startswith("denoise_fn.") else v)
                         for k, v in rec["state_dict"].items()}
    ora = oracle_model(rec)
    loss, _ = ora.p_losses(rec["x0"], rec["y"], None, rec["t"], rec["noise"])
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in rec["state_dict"].items() if k.startswith("denoise_fn.")}


@pytest.mark.parametrize("name", CASES)
def test_loss_and_all_parameter_gradients(dev, name):
    rec = load_case(name)
    loss_ref, g_ref = _oracle_grads(rec)
    m = build(rec, dev).train()
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else rec["y"].to(dev)
    loss, log = m.p_losses(rec["x0"].to(dev), rec["y"].to(dev), ctx, rec["t"].to(dev), rec["noise"].to(dev))
    assert loss.requires_grad and "x0_recon" in log
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    # A conv bias that feeds a GroupNorm has a (mathematically) ~zero gradient: its reference value is rounding noise.
    # Errors are therefore measured against max(|g_ref|_max, 1e-3 * largest gradient magnitude in the model).
    gmax = max(float(v.abs().max()) for v in g_ref.values())
    errs = []
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        errs.append((float((p.grad.cpu() - ref).abs().max()) / scale, k, float(ref.abs().max())))
    errs.sort(reverse=True)
    print(f"{name}: loss {float(loss):.6f} (ref {loss_ref:.6f}); gmax {gmax:.3e}; worst: " +
          "; ".join(f"{k} {e:.2e} (|g| {g:.1e})" for e, k, g in errs[:4]))
    assert errs[0][0] < GRAD_TOL, errs[0]


def test_skip_projection_on_a_second_stream_training_is_bit_equal(dev):
    """UNetModel.side_stream_train: the 1x1 skip projections of a training plan -- their forward launch, and in the gradient plan their
    weight gradient + data gradient on a workspace of their own -- run on a second stream, forked at the top of the block and joined
    before the launch that consumes them.  Same kernels: the output and every parameter gradient are bit-equal to the one-stream plan,
    three micro-steps in a row (a missing join shows up as a difference sooner or later).  A 64-channel UNet on 16 images of 16x16 with
    the band forced open and the projections on the bf16x3 GEMM, as at full size (the direct kernels own a shared workspace and stay on
    stream one)."""
    import bbdm_amd
    from fixture_weights import synth_weights
    up = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=True, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(16, 4, 16, 16, generator=g).to(dev)
    t = (torch.arange(16) * 5 + 2).to(dev)
    dout = torch.randn(16, 4, 16, 16, generator=g).to(dev)
    runs = {}
    for band in (0, 1 << 62):
        m = bbdm_amd.unet.UNetModel(**up)
        sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 44)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        m.side_stream_min_macs, m.side_stream_max_macs, m.side_stream_max_pixels, m.side_stream_train = 0, band, 1 << 30, bool(band)
        m.bf3_min_tiles, m.side_stream_wgrad_min_macs = 1, 0
        outs = []
        for rep in range(3):
            m.zero_grad(set_to_none=True)
            out = m(x, timesteps=t, context=None)
            (out * dout).sum().backward()
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
        plan = next(p for p in m._plans.values() if p.training)
        assert bool(plan._side_ranges) == bool(band) and bool(plan._bside_ranges) == bool(band), (len(plan._side_ranges), len(plan._bside_ranges))
        kinds = set()
        for k0, k1, kj in plan._bside_ranges:
            names = [str(n) for n, _ in plan.bops[k0:k1]]
            assert k0 < k1 <= kj and str(plan.bops[kj][0]) == "bbdm_groupnorm_bwd_f32"
            assert names == ["bbdm_conv_wgrad_f32", "bbdm_conv1x1_bf3_f32"] or names[:2] == ["bbdm_winograd_dy_transform_bf3p_f32", "bbdm_gemm_bf3p_tn_f32"], names
            kinds.add(names[0])
        assert not band or len(kinds) == 2, kinds        # both the projections' gradients and the 3x3 layers' weight-gradient chains
        runs[band] = outs
    for rep in range(3):
        oa, ga = runs[0][rep]
        ob, gb = runs[1 << 62][rep]
        assert torch.equal(oa, ob), rep
        bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
        assert not bad, (rep, bad[:6])


@pytest.mark.parametrize("name", ["tiny_concat", "tiny_xattn"])
def test_gradient_accumulation_and_input_grad(dev, name):
    """Two backward passes accumulate into .grad (accumulate_grad_batches, BaseRunner.py:412-417); d loss / d context
    reaches a conditioning input that requires grad (trainable SpatialRescaler case).  tiny_xattn: the to_k / to_v weights of
    the cross-attention act on a 3-channel context padded to 4 -- their gradients are cut out of a padded scratch tensor by the
    backward SEGMENT that owns them (a copy deferred to the last segment handed autograd a still-zero view: the second pass then
    accumulated 1x instead of 2x, and DDP reduced zeros)."""
    rec = load_case(name)
    m = build(rec, dev).train()
    x0, y, t, nz = (rec[k].to(dev) for k in ("x0", "y", "t", "noise"))
    ctx = y.clone().requires_grad_()
    loss, _ = m.p_losses(x0, y, ctx, t, nz)
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert ctx.grad is not None and float(ctx.grad.abs().max()) > 0
    # oracle for d/d context
    sd = rec["state_dict"]
    ora = oracle_model(rec)
    c_ref = rec["y"].clone().requires_grad_()
    x_t, target = O.q_sample(ora.bufs, rec["x0"], rec["y"], rec["t"], rec["noise"], ora.objective)
    l_ref = O.bb_loss(target, ora.denoise(x_t, rec["t"], c_ref), ora.loss_type)
    l_ref.backward()
    assert rel_err(ctx.grad.cpu(), c_ref.grad) < GRAD_TOL
    loss2, _ = m.p_losses(x0, y, y, t, nz)
    loss2.backward()
    for k, p in m.named_parameters():
        assert rel_err(p.grad, 2 * g1[k]) < 1e-5, k


@pytest.mark.parametrize("name", ["tiny_nocond", "tiny_xattn"])
def test_accumulation_in_place_matches_autograd(dev, name):
    """dist_utils.accumulation_sync lets the UNet add a micro-step's gradients to the ``.grad`` tensors itself (one add per contiguous
    run of the flat gradient buffer instead of one AccumulateGrad launch per parameter): same bits as autograd's accumulation over
    three micro-steps and ``.grad`` keeps its storage -- for this package's own bare model (and a torch DDP under ``no_sync``:
    tests/test_dist_cpu.py).  Everything that could be WATCHING gradients arrive keeps autograd's path (round-5 advisor finding): a
    wrapper that is not torch's DistributedDataParallel (with or without ``no_sync``), and -- inside a bare model -- a parameter with a
    tensor hook or a post-accumulate-grad hook, whose hook must fire on every micro-step."""
    import contextlib
    from bbdm_amd import dist_utils
    rec = load_case(name)
    x0, y, t, nz = (rec[k].to(dev) for k in ("x0", "y", "t", "noise"))
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else y
    seen_in_place = []

    def run(wrap, prepare=None):
        m = build(rec, dev).train()
        if prepare is not None:
            prepare(m)
        net = wrap(m)
        ptrs = None
        for step in (1, 2, 3):
            with dist_utils.accumulation_sync(net, step, 4) if wrap is not _plain else contextlib.nullcontext():
                seen_in_place.append(bool(m.denoise_fn.grad_in_place))
                loss, _ = m.p_losses(x0 * (1.0 / step), y, ctx, t, nz)
                loss.backward()
            if step == 1:
                ptrs = {k: p.grad.data_ptr() for k, p in m.named_parameters()}
        if dev.type == "cuda":
            torch.cuda.synchronize()
        return m, ptrs

    _plain = lambda m: m
    want, _ = run(_plain)
    assert seen_in_place == [False] * 3

    del seen_in_place[:]
    got, ptrs = run(lambda m: _Bare(m))             # the bare model: every micro-step may accumulate in place
    assert seen_in_place == [True] * 3
    for (k, p), (_, q) in zip(got.named_parameters(), want.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
        assert p.grad.data_ptr() == ptrs[k], k                 # accumulated in place: .grad never re-pointed

    class _Wrapper:                                 # NOT torch DDP: its gradient handling is unknown -> autograd's path on every micro-step
        def __init__(self, m): self.m = m
        def modules(self): return self.m.modules()
    adds = []

    class _WrapperWithNoSync(_Wrapper):             # ... its no_sync is still honoured on the non-boundary micro-steps
        @contextlib.contextmanager
        def no_sync(self):
            adds.append(1)
            yield
    for wrap in (_Wrapper, _WrapperWithNoSync):
        del seen_in_place[:]
        got2, _ = run(wrap)
        assert seen_in_place == [False] * 3, wrap
        for (k, p), (_, q) in zip(got2.named_parameters(), want.named_parameters()):
            assert torch.equal(p.grad, q.grad), k
        assert not got2.denoise_fn.grad_in_place
    assert len(adds) == 3                          # micro-steps 1..3 of 4 are non-boundary

    # observers on individual parameters of a bare model: their segment leaves the in-place path, the hooks fire every micro-step
    fired = {"post": 0, "tensor": 0}

    def prepare(m):
        ps = list(m.denoise_fn.parameters())
        ps[3].register_post_accumulate_grad_hook(lambda p: fired.__setitem__("post", fired["post"] + 1))
        ps[-2].register_hook(lambda g: fired.__setitem__("tensor", fired["tensor"] + 1))
    got3, _ = run(lambda m: _Bare(m), prepare)
    assert fired == {"post": 3, "tensor": 3}, fired
    for (k, p), (_, q) in zip(got3.named_parameters(), want.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
    assert not got.denoise_fn.grad_in_place and not got3.denoise_fn.grad_in_place


def _Bare(m):
    """(the model object itself: accumulation_sync recognises this package's own classes)"""
    return m


def test_gradients_handed_out_are_never_overwritten(dev):
    """The parameter gradients are views of a recycled flat buffer: tensors returned by torch.autograd.grad, or a .grad the caller
    keeps across zero_grad(set_to_none=True), must survive the next backward pass (gradient penalties, logging, manual
    accumulation)."""
    rec = load_case("tiny_nocond")
    m = build(rec, dev).train()
    nb = None if dev.type == "cuda" else 1              # (one image on the CPU-emulated kernels: the property is batch-independent)
    x0, y, t, nz = (rec[k][:nb].to(dev) for k in ("x0", "y", "t", "noise"))
    params = [p for p in m.parameters() if p.requires_grad]
    loss, _ = m.p_losses(x0, y, None, t, nz)
    ga = torch.autograd.grad(loss, params)
    snap = [g.clone() for g in ga]
    loss, _ = m.p_losses(x0 * 0.5, y, None, t, nz)          # different data: different gradients
    gb = torch.autograd.grad(loss, params)
    assert all(torch.equal(a, s) for a, s in zip(ga, snap))
    assert any(not torch.equal(a, b) for a, b in zip(ga, gb))
    loss, _ = m.p_losses(x0, y, None, t, nz)
    loss.backward()
    kept = [p.grad for p in params]
    snap = [g.clone() for g in kept]
    m.zero_grad(set_to_none=True)
    for scale in (0.5, 0.25, 2.0):                            # more passes than there are persistent buffers
        loss, _ = m.p_losses(x0 * scale, y, None, t, nz)
        loss.backward()
        m.zero_grad(set_to_none=True)
    assert all(torch.equal(a, s) for a, s in zip(kept, snap))


def test_context_gradient_through_cross_attention(dev):
    """use_spatial_transformer: the context reaches the UNet twice -- concatenated to the input (openaimodel.py:741-742) and as
    the cross-attention keys / values of every SpatialTransformer (attention.py:174-176); d loss / d context sums both."""
    rec = load_case("tiny_xattn")
    m = build(rec, dev).train()
    x0, y, t, nz = (rec[k].to(dev) for k in ("x0", "y", "t", "noise"))
    ctx = y.clone().requires_grad_()
    loss, _ = m.p_losses(x0, y, ctx, t, nz)
    loss.backward()
    ora = oracle_model(rec)
    c_ref = rec["y"].clone().requires_grad_()
    x_t, target = O.q_sample(ora.bufs, rec["x0"], rec["y"], rec["t"], rec["noise"], ora.objective)
    l_ref = O.bb_loss(target, ora.denoise(x_t, rec["t"], c_ref), ora.loss_type)
    l_ref.backward()
    assert abs(float(loss.detach()) - float(l_ref.detach())) < 1e-5 * max(1.0, abs(float(l_ref.detach())))
    assert rel_err(ctx.grad.cpu(), c_ref.grad) < GRAD_TOL


@pytest.mark.parametrize("fused", [False, True])
def test_adam_steps_follow_the_oracle(dev, fused):
    """Five Adam steps on one batch: the loss curve tracks an identical run of the oracle on the CPU -- with torch.optim.Adam and
    with bbdm_amd.optim.FusedAdam, whose kernel rewrites the parameters outside autograd: the forward of the next step must see
    the new weights (packed conv copies are keyed on the parameter's version counter, which FusedAdam bumps)."""
    from bbdm_amd.optim import FusedAdam
    rec = load_case("tiny_nocond")
    m = build(rec, dev).train()
    opt = (FusedAdam if fused else torch.optim.Adam)(m.get_parameters(), lr=1e-4, betas=(0.9, 0.999))
    rec_o = dict(rec)
    rec_o["state_dict"] = {k: (v.clone().requires_grad_() if k.startswith("denoise_fn.") else v)
                           for k, v in rec["state_dict"].items()}
    ora = oracle_model(rec_o)
    opt_o = torch.optim.Adam([v for k, v in rec_o["state_dict"].items() if k.startswith("denoise_fn.")], lr=1e-4,
                             betas=(0.9, 0.999))
    nb = None if dev.type == "cuda" else 1              # (one image, three steps on the CPU-emulated kernels)
    x0, y, t, nz = (rec[k][:nb] for k in ("x0", "y", "t", "noise"))
    a, b = [], []
    for _ in range(5 if dev.type == "cuda" else 3):
        opt.zero_grad()
        loss, _ = m.p_losses(x0.to(dev), y.to(dev), None, t.to(dev), nz.to(dev))
        loss.backward()
        opt.step()
        a.append(float(loss))
        opt_o.zero_grad()
        lo, _ = ora.p_losses(x0, y, None, t, nz)
        lo.backward()
        opt_o.step()
        b.append(float(lo))
    print("loss curve gpu:", a, "oracle:", b)
    assert a[-1] < a[0]
    for u, v in zip(a, b):
        assert abs(u - v) < 2e-4 * max(1.0, abs(v))


def test_ddp_wrapping_single_rank(dev):
    """runners/BaseRunner.py:76 wraps the model in DistributedDataParallel; with world size 1 over RCCL the reducer
    hooks must fire for every parameter of the single custom autograd node."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        rec = load_case("tiny_concat")
        m = build(rec, dev).train()
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], output_device=0)
        torch.manual_seed(0)
        loss, _ = ddp(rec["x0"].to(dev), rec["y"].to(dev))
        loss.backward()
        assert all(p.grad is not None for p in m.get_parameters())
        loss, _ = ddp(rec["x0"].to(dev), rec["y"].to(dev))       # second iteration (reducer re-arm)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def test_batch_larger_than_64(dev):
    """The embedding-path kernels take <= 64 rows per call: batches above that are chunked on the host, in the forward
    (time MLP, FiLM projections) and in the backward (per-chunk weight gradients are summed)."""
    rec = load_case("tiny_nocond")
    g = torch.Generator().manual_seed(3)
    N = 70
    x0 = torch.randn(N, 8, 8, 8, generator=g)
    y = torch.randn(N, 8, 8, 8, generator=g)
    t = torch.randint(0, 50, (N,), generator=g)
    nz = torch.randn(N, 8, 8, 8, generator=g)
    rec_o = dict(rec)
    rec_o["state_dict"] = {k: (v.clone().requires_grad_() if k.startswith("denoise_fn.") else v)
                           for k, v in rec["state_dict"].items()}
    ora = oracle_model(rec_o)
    lo, _ = ora.p_losses(x0, y, None, t, nz)
    lo.backward()
    m = build(rec, dev).train()
    loss, _ = m.p_losses(x0.to(dev), y.to(dev), None, t.to(dev), nz.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-5
    gmax = max(float(v.grad.abs().max()) for k, v in rec_o["state_dict"].items() if k.startswith("denoise_fn."))
    for k, p in m.named_parameters():
        ref = rec_o["state_dict"][k].grad
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        assert float((p.grad.cpu() - ref).abs().max()) / scale < GRAD_TOL, k
    with torch.no_grad():
        out = m.denoise_fn(x0.to(dev), timesteps=t.to(dev), context=None)
        ref_out = ora.denoise(x0, t, None)
    assert rel_err(out.cpu(), ref_out.detach()) < 1e-4


def test_gradients_through_winograd_layers(dev):
    """A UNet wide and large enough (128 / 256 channels, batch 16, 32x32) for the forward AND the data-gradient
    convolutions to take the Winograd F(4x4,3x3) path (bbdm_amd.unet.winograd_tile): loss and every parameter gradient
    against autograd on the CPU oracle, with the Winograd path and -- same weights, same batch -- with the direct kernel
    only.  (The golden-fixture models above are too narrow for the Winograd path.)  Kept last in the file on purpose."""
    import bbdm_amd
    from fixture_weights import synth_weights
    up = dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=1,
              attention_resolutions=(2,), channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=8,
              num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True, use_spatial_transformer=False,
              context_dim=None, condition_key="nocond")
    bb = dict(mt_type="linear", objective="grad", loss_type="l1", skip_sample=True, sample_type="linear",
              sample_step=50, num_timesteps=1000, eta=1.0, max_var=1.0)
    m = bbdm_amd.BrownianBridgeModel(_ns({"BB": {"params": dict(bb, UNetParams=up)}}))
    shapes = [(k, tuple(v.shape)) for k, v in m.denoise_fn.state_dict().items()]
    sd = synth_weights(shapes, 4242, w_std=0.02)
    m.denoise_fn.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(99)
    N = 16
    x0 = torch.randn(N, 3, 32, 32, generator=g).clamp(-1, 1)
    y = torch.randn(N, 3, 32, 32, generator=g).clamp(-1, 1)
    t = torch.randint(0, 1000, (N,), generator=g)
    nz = torch.randn(N, 3, 32, 32, generator=g)
    # oracle
    osd = {"denoise_fn." + k: v.clone().requires_grad_() for k, v in sd.items()}
    ora = O.OracleBBDM(osd, O.UNetSpec(**up), **bb)
    lo, _ = ora.p_losses(x0, y, None, t, nz)
    lo.backward()
    g_ref = {k[len("denoise_fn."):]: v.grad for k, v in osd.items()}
    gmax = max(float(v.abs().max()) for v in g_ref.values())
    m = m.to(dev).train()
    for wino in (6, 4, 0):
        m.denoise_fn.winograd = wino
        m.zero_grad(set_to_none=True)
        loss, _ = m.p_losses(x0.to(dev), y.to(dev), None, t.to(dev), nz.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        plan = m.denoise_fn._plan_for(x0.to(dev), True)
        fw = sum(n == "bbdm_winograd_gemm_f32" for n, _ in plan.ops)
        bw = sum(n == "bbdm_winograd_gemm_f32" for n, _ in plan.bops)
        assert (fw > 0 and bw > 0) if wino else (fw == 0 and bw == 0)
        errs = []
        for k, p in m.denoise_fn.named_parameters():
            ref = g_ref[k]
            scale = max(float(ref.abs().max()), 1e-3 * gmax)
            errs.append((float((p.grad.cpu() - ref).abs().max()) / scale, k))
        errs.sort(reverse=True)
        print(f"winograd={wino}: {fw} forward + {bw} data-gradient Winograd layers; loss {float(loss.detach()):.6f} "
              f"(ref {float(lo.detach()):.6f}); worst gradient errors: " + "; ".join(f"{k} {e:.2e}" for e, k in errs[:3]))
        assert abs(float(loss.detach()) - float(lo.detach())) < 1e-5 * max(1.0, abs(float(lo.detach())))
        assert errs[0][0] < GRAD_TOL, errs[0]


def test_upsample_conv_folded_into_the_winograd_input_transform(dev):
    """`resblock_updown=False` models resample with `Upsample` (nearest x2 + 3x3 conv) / `Downsample` (stride-2 conv):
    in inference plans the nearest x2 is folded into the Winograd input transform of the conv (the 4x tensor is never
    written).  Forward parity against the CPU oracle, Winograd path and direct-only, on a model wide enough to take it."""
    import bbdm_amd
    from fixture_weights import synth_weights
    up = dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=1,
              attention_resolutions=(2,), channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=8,
              num_head_channels=64, use_scale_shift_norm=True, resblock_updown=False, use_spatial_transformer=False,
              context_dim=None, condition_key="nocond")
    bb = dict(mt_type="linear", objective="grad", loss_type="l1", skip_sample=True, sample_type="linear",
              sample_step=50, num_timesteps=1000, eta=1.0, max_var=1.0)
    m = bbdm_amd.BrownianBridgeModel(_ns({"BB": {"params": dict(bb, UNetParams=up)}}))
    shapes = [(k, tuple(v.shape)) for k, v in m.denoise_fn.state_dict().items()]
    sd = synth_weights(shapes, 515, w_std=0.02)
    m.denoise_fn.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(17)
    N = 16
    x = torch.randn(N, 3, 32, 32, generator=g).clamp(-1, 1)
    t = torch.randint(0, 1000, (N,), generator=g)
    ora = O.OracleBBDM({"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**up), **bb)
    with torch.no_grad():
        ref = ora.denoise(x, t, None)
    m = m.to(dev).eval()
    for wino, phases in ((4, True), (4, False), (0, True)):
        m.denoise_fn.winograd, m.denoise_fn.upsample_phases = wino, phases
        with torch.no_grad():
            out = m.denoise_fn(x.to(dev), timesteps=t.to(dev), context=None)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        plan = m.denoise_fn._plan_for(x.to(dev), False)
        # round 3: the conv runs as its four phase filters on the LOW-resolution tensor (output transform scatters the phases,
        # flag 8); upsample_phases = False: the round-2 form, index shift inside the input transform of the upsampled size
        folded = sum(n == "bbdm_winograd_input_f32" and a[8] == 1 for n, a in plan.ops)
        as_phases = sum(n == "bbdm_winograd_output_f32" and a[7] == 8 for n, a in plan.ops)
        assert (folded, as_phases) == ((0, 1) if (wino and phases) else (1, 0) if wino else (0, 0))
        e = rel_err(out.cpu(), ref)
        print(f"winograd={wino} phases={phases}: {folded} folded / {as_phases} as phase filters; UNet forward rel err {e:.2e}")
        assert e < 1e-4
