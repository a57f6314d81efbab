"""Model-level paths on the CPU-emulated kernels (tools/hipemu): the execution plan, the fused bridge update, the
autograd node and its backward plan run end to end on the golden fixtures of the real reference -- the same assertions
as tests/test_model_gpu.py / test_training_gpu.py, before any GPU time is spent.  Not a performance path."""
import os
import pytest
import torch

import test_model_gpu as M
import test_training_gpu as T
from emu_backend import emulated_backend
from fixtures import load_case, parity_err, rel_err

CPU = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emulated_backend() as emu:
        yield emu


def _build(rec, train=False):
    m = T.build(rec, CPU)
    m.denoise_fn.hip_graph = False            # hipGraph capture needs a real device
    return m.train() if train else m.eval()


@pytest.mark.parametrize("name", ["tiny_concat", "tiny_ysubx", "tiny_xattn"])
def test_unet_forward_and_p_sample_match_reference_golden(name):
    rec = load_case(name)
    m = _build(rec)
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else rec["y"]
    with torch.no_grad():
        out = m.denoise_fn(rec["x0"], timesteps=rec["t"], context=ctx)
    assert parity_err(out, rec["unet_out"]) < M.STEP_TOL
    # the plan that launches its 1x1 skip projections FIRST (on a GPU: on the graph's second stream, UNetModel.side_stream_*): the op
    # order the emulator runs is a valid serialisation of it -- same bits
    fn = m.denoise_fn
    fn.side_stream_min_macs, fn.side_stream_max_macs, fn.side_stream_max_pixels = 0, 1 << 62, 1 << 30
    with torch.no_grad():
        out2 = fn(rec["x0"], timesteps=rec["t"], context=ctx)
    plan2 = [p for p in fn._plans.values() if p._side_ranges]
    assert len(plan2) == 1 and torch.equal(out2, out)
    fn.side_stream_max_macs = 0
    eps = rec["p_eps"]
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: eps
    try:
        for clip, i, a_ref, b_ref in rec["p_out"][:2]:
            a, b = m.p_sample(rec["p_x_t"], rec["y"], ctx, i, clip_denoised=clip)
            assert parity_err(a, a_ref) < M.STEP_TOL and parity_err(b, b_ref) < M.STEP_TOL, (clip, i)
    finally:
        torch.randn_like = orig


@pytest.mark.parametrize("name", ["tiny_nocond", "tiny_xattn"])
def test_loss_and_all_parameter_gradients(name):
    rec = load_case(name)
    loss_ref, g_ref = T._oracle_grads(rec)
    m = _build(rec, train=True)
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else rec["y"]
    loss, log = m.p_losses(rec["x0"], rec["y"], ctx, rec["t"], rec["noise"])
    loss.backward()
    assert abs(float(loss.detach()) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    gmax = max(float(v.abs().max()) for v in g_ref.values())
    for k, p in m.named_parameters():
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        assert float((p.grad - ref).abs().max()) / scale < T.GRAD_TOL, k


def test_context_gradient_through_cross_attention():
    T.test_context_gradient_through_cross_attention(CPU)


@pytest.mark.parametrize("name", ["tiny_concat", "tiny_xattn"])
def test_gradient_accumulation_and_input_grad(name):
    T.test_gradient_accumulation_and_input_grad(CPU, name)


def test_gradients_handed_out_are_never_overwritten():
    T.test_gradients_handed_out_are_never_overwritten(CPU)


def test_fused_adam_steps_follow_the_oracle():
    T.test_adam_steps_follow_the_oracle(CPU, True)


def test_groupnorm_statistics_fused_into_the_producers():
    """A UNet wide enough (128 channels -> 4 per group) for the conv epilogues to accumulate the GroupNorm statistics of
    their outputs (block outputs feeding the next block AND, through the concat, an output block): same result as with the
    stand-alone statistics pass, and both match the oracle."""
    import argparse
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    up = dict(image_size=16, in_channels=4, model_channels=128, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1,), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=True, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 31)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(16, 4, 16, 16, generator=g)          # 16 images: enough 4x4 tiles for the Winograd path at 16x16
    t = torch.arange(16) * 3 + 5
    outs = {}
    for fuse in (True, False):
        m.fuse_stats = fuse
        with torch.no_grad():
            outs[fuse] = m(x, timesteps=t, context=None).clone()
        plan = m._plan_for(x, False)
        n_alone = sum(str(n) == "bbdm_groupnorm_stats_f32" for n, _ in plan.ops)
        assert (plan.fused_stats > 0) == fuse and (n_alone == 0 or not fuse or plan.fused_stats > 0)
        if fuse:
            assert plan.fused_stats >= 3, (plan.fused_stats, n_alone)
    ref = O.unet_forward(sd, O.UNetSpec(**up), x, t, None)
    assert parity_err(outs[True], outs[False]) < 1e-6
    assert parity_err(outs[True], ref) < M.STEP_TOL


def test_sampling_plan_on_the_presplit_gemm_is_bit_equal(monkeypatch):
    """Inference plan with the Winograd layers on csrc/gemm_bf3p.hip (input transform writes the three bf16 planes, LDS-DMA GEMM):
    bit-equal to the same plan on csrc/gemm_bf3.hip (fp32 V split while staged), and within the step tolerance of the oracle."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    monkeypatch.setattr(bbdm_amd.unet, "winograd_tile",
                        lambda N, H, W, cin, cout, max_m=6, small=True, allow8=False: 4 if (max_m >= 4 and cin >= 64 and cout >= 64 and H % 4 == 0) else 0)
    up = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=False, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 43)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.gemm_h2 = False              # (the fp16-pair planes are a different product: test_sampling_plan_on_the_fp16_pair_planes)
    m.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 4, 16, 16, generator=g)
    t = torch.arange(16) * 7 + 1
    outs = {}
    for p in (True, False):
        m.gemm_bf3p = p
        with torch.no_grad():
            outs[p] = m(x, timesteps=t, context=None).clone()
        plan = m._plan_for(x, False)
        entries = [getattr(n, "entry", str(n)) for n, _ in plan.ops]
        assert ("bbdm_winograd_gemm_bf3p_f32" in entries) == p and ("bbdm_winograd_input_bf3p_f32" in entries) == p
        assert ("bbdm_winograd_gemm_bf3_f32" in entries) == (not p)
    assert torch.equal(outs[True], outs[False])
    ref = O.unet_forward(sd, O.UNetSpec(**up), x, t, None)
    assert parity_err(outs[True], ref) < M.STEP_TOL


def test_sampling_plan_on_the_fp16_pair_planes(monkeypatch):
    """Inference plan with the GroupNorm-fed Winograd layers on two fp16 planes per operand (csrc/h2_split.h; UNetModel.gemm_h2, the
    default): every ResBlock convolution takes the h2 entry points with a bound slot of its own, the bounds launch fills them from
    gamma / beta / the FiLM vector, and the step stays within the tolerance of the oracle -- also when the weights are 50x larger than the
    initialisation (the scale follows the bound: nothing overflows fp16) and when an input pixel is a 1000x outlier."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    monkeypatch.setattr(bbdm_amd.unet, "winograd_tile",
                        lambda N, H, W, cin, cout, max_m=6, small=True, allow8=False: 4 if (max_m >= 4 and cin >= 64 and cout >= 64 and H % 4 == 0) else 0)
    up = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=True, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 44)
    m.hip_graph = False
    m.eval()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(16, 4, 16, 16, generator=g)
    t = torch.arange(16) * 7 + 1
    for case in ("plain", "gain50", "outlier"):
        sd_c = {k: (v * 50 if case == "gain50" and (".in_layers.2." in k or ".out_layers.3." in k or "norm" in k or ".in_layers.0." in k or ".out_layers.0." in k) and k.endswith("weight") else v) for k, v in sd.items()}
        m.load_state_dict(sd_c, strict=True)
        xc = x.clone()
        if case == "outlier":
            xc[:, :, 3, 5] *= 1000.0
        with torch.no_grad():
            out = m(xc, timesteps=t, context=None).clone()
        plan = m._plan_for(xc, False)
        entries = [getattr(n, "entry", str(n)) for n, _ in plan.ops]
        n_h2 = sum(e in ("bbdm_winograd_gemm_h2p_f32", "bbdm_winograd_gemm_h2p_splitk_f32") for e in entries)
        assert n_h2 >= 6 and n_h2 == len([1 for e in entries if e.startswith("bbdm_winograd_input_h2p")]) and len(plan._h2_layers) >= n_h2
        bounds = plan._h2_bounds.t
        assert bool(torch.isfinite(bounds).all()) and float(bounds[:len(plan._h2_layers)].min()) > 0
        assert bool(torch.isfinite(out).all())
        ref = O.unet_forward(sd_c, O.UNetSpec(**up), xc, t, None)
        assert parity_err(out, ref) < M.STEP_TOL, (case, parity_err(out, ref))


def test_upsampling_resblock_resamples_inside_the_transforms(monkeypatch):
    """resblock_updown: an up-sampling ResBlock of an inference plan emits NO resampling pass -- GroupNorm -> SiLU -> nearest x2 is the
    index shift of the first conv's Winograd input transform and x_upd(x) the out conv's residual read at [h/2][w/2]
    (BBDM_CONV_RES_UPSAMPLE) -- and computes what the explicit passes compute (openaimodel.py:259-264), which is what the oracle does."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    monkeypatch.setattr(bbdm_amd.unet, "winograd_tile",
                        lambda N, H, W, cin, cout, max_m=6, small=True, allow8=False: 4 if (max_m >= 4 and cin >= 64 and cout >= 64 and H % 4 == 0) else 0)
    up = dict(image_size=8, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=True, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 47)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(16, 4, 8, 8, generator=g)
    t = torch.arange(16) * 11 + 3
    outs = {}
    for fold in (True, False):
        m.winograd_fuse_groupnorm = fold
        with torch.no_grad():
            outs[fold] = m(x, timesteps=t, context=None).clone()
        plan = m._plan_for(x, False)
        resamplers = [a for n, a in plan.ops if str(n) == "bbdm_groupnorm_apply_f32" and a[-1] == 2]
        up_reads = [a for n, a in plan.ops if str(n) == "bbdm_winograd_output_f32" and a[7] == 4]
        assert (len(resamplers) == 0 and len(up_reads) == 1) if fold else (len(resamplers) == 2 and not up_reads)
        # ... and the first conv of the folded block runs as the four phase filters of conv3x3(nearest x2 (.)) on the LOW-resolution tensor
        phase_convs = [a for n, a in plan.ops if str(n) == "bbdm_winograd_output_f32" and a[7] == 8]
        assert len(phase_convs) == (1 if fold else 0)
    assert parity_err(outs[True], outs[False]) < 2e-5
    m.winograd_fuse_groupnorm, m.upsample_phases = True, False         # the round-2 form: input transform of the upsampled tensor
    with torch.no_grad():
        out_up = m(x, timesteps=t, context=None).clone()
    plan = m._plan_for(x, False)
    assert not any(str(n) == "bbdm_winograd_output_f32" and a[7] == 8 for n, a in plan.ops)
    assert any(str(n) == "bbdm_winograd_input_f32" and a[8] == 1 for n, a in plan.ops)
    assert parity_err(outs[True], out_up) < 2e-5
    ref = O.unet_forward(sd, O.UNetSpec(**up), x, t, None)
    assert parity_err(outs[True], ref) < M.STEP_TOL


def test_upsampling_conv_as_f72_phase_filters(monkeypatch):
    """The up-sampling ResBlock's first conv as four 2 x 2 phase filters on F(7x7, 2x2) (round 5; unet.phase_filter_tile chooses it where
    the layer earns the 8-point transform -- forced here on an 8x8 model): same result as the F(m x m, 3x3) phase filters and as the
    oracle's explicit upsample + conv (openaimodel.py:259-264)."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    wt = lambda N, H, W, cin, cout, max_m=6, small=True, allow8=False: 4 if (max_m >= 4 and cin >= 64 and cout >= 64 and H % 4 == 0) else 0
    monkeypatch.setattr(bbdm_amd.unet, "winograd_tile", wt)
    monkeypatch.setattr(bbdm_amd.unet, "phase_filter_tile",
                        lambda N, H, W, cin, cout4, max_m, small, f72=True:
                        7 if (f72 and (cout4 // 4) % 128 == 0 and cin % 16 == 0) else wt(N, H, W, cin, cout4, max_m, small))
    up = dict(image_size=8, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1, 2), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=True, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 48)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.eval()
    g = torch.Generator().manual_seed(10)
    x = torch.randn(6, 4, 8, 8, generator=g)
    t = torch.arange(6) * 17 + 5
    outs = {}
    for f72 in (True, False):
        m.upsample_f72 = f72
        with torch.no_grad():
            outs[f72] = m(x, timesteps=t, context=None).clone()
        plan = m._plan_for(x, False)
        tiles7 = [a for n, a in plan.ops if str(n) == "bbdm_winograd_output_f32" and a[0] == 7 and a[7] == 8]
        assert len(tiles7) == (1 if f72 else 0)
    assert parity_err(outs[True], outs[False]) < 2e-5
    ref = O.unet_forward(sd, O.UNetSpec(**up), x, t, None)
    assert parity_err(outs[True], ref) < M.STEP_TOL


def test_weight_gradients_in_the_winograd_domain(monkeypatch):
    """Training plan of a UNet with enough tiles (16 images of 16x16 = 256 4x4 tiles, 64 channels) for the 3x3 layers' weight
    gradients to take the Winograd-domain path (csrc/winograd_wgrad.hip), with the 1x1 skip convolutions (forward and data
    gradient) on the bf16x3 GEMM as at full size: every parameter gradient against the oracle's autograd."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    # the forward's tile rule wants >= 128 channels (4x the emulation time): let the 64-channel layers take F(4x4) too, so that
    # the training forward keeps their V and the gradient plan runs the staged form (dY transform -> TN GEMM -> finish) on it
    monkeypatch.setattr(bbdm_amd.unet, "winograd_tile",
                        lambda N, H, W, cin, cout, max_m=6, small=True, allow8=False: 4 if (max_m >= 4 and cin >= 64 and cout >= 64 and H % 4 == 0) else 0)
    up = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1,), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=False, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 41)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.bf3_min_tiles = 1        # also drive the 1x1 skip convolutions (forward and data gradient) through the bf16x3 GEMM
    m.train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 4, 16, 16, generator=g)
    t = torch.arange(16) * 5 + 2
    dout = torch.randn(16, 4, 16, 16, generator=g)
    (m(x, timesteps=t, context=None) * dout).sum().backward()
    plan = m._plan_for(x, True)
    # (layers whose forward ran the same Winograd tile keep their V: the gradient plan then holds the stages, not the chained entry)
    assert sum(str(n) in ("bbdm_conv3x3_winograd_wgrad_f32", "bbdm_winograd_wgrad_finish_f32", "bbdm_winograd_wgrad_finish_bias_f32") for n, _ in plan.bops) >= 4
    # ... as TRANSPOSED bf16 planes, contracted by the bf16x3 GEMM (csrc/gemm_bf3p.hip: the tiles are its K loop)
    assert sum(str(n) == "bbdm_gemm_bf3p_tn_f32" for n, _ in plan.bops) >= 1
    assert any(getattr(n, "entry", "") in ("bbdm_winograd_input_bf3p_tr_f32", "bbdm_winograd_input_h2p_tr_f32", "bbdm_winograd_input_h2p_tr2_f32") for n, _ in plan.ops)
    assert len(plan._fused_train) >= 2          # ... and those layers' GN -> SiLU input was folded into the transform, not materialised
    assert sum(str(n) == "bbdm_conv1x1_bf3_f32" for n, _ in plan.bops) >= 1 and \
        sum(str(n) == "bbdm_conv1x1_bf3_f32" for n, _ in plan.ops) >= 1
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items()}
    (O.unet_forward(sdg, O.UNetSpec(**up), x, t, None) * dout).sum().backward()
    gmax = max(float(v.grad.abs().max()) for v in sdg.values())
    for k, p in m.named_parameters():
        ref = sdg[k].grad
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        assert float((p.grad - ref).abs().max()) / scale < T.GRAD_TOL, k


def test_training_plan_on_f8_tiles(monkeypatch):
    """UNetModel.winograd_train8 = 2 (the default): forward, data gradient and Winograd-domain weight gradient of a training plan on
    F(8x8, 3x3) -- the transposed planes of V from the m = 8 input transform, A dY A^T on ten points with the tile sums for the bias
    gradient, G^T dU G in fp64 -- forced onto a 128-channel 16x16 model (the real rule wants >= 512 tiles): every parameter gradient
    against the oracle's autograd; winograd_train8 = 0 plans the same model without the tile."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    monkeypatch.setattr(bbdm_amd.unet, "_tile8_ok", lambda N, H, W, cin, cout, min_tiles=512: cin >= 128 and cin % 32 == 0 and cout % 128 == 0 and H % 8 == 0)
    up = dict(image_size=16, in_channels=4, model_channels=128, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1,), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=False, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 43)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 16, 16, generator=g)
    t = torch.arange(2) * 31 + 2
    dout = torch.randn(2, 4, 16, 16, generator=g)
    (m(x, timesteps=t, context=None) * dout).sum().backward()
    plan = m._plan_for(x, True)
    f8 = [a for n, a in plan.ops if str(n) == "bbdm_winograd_gemm_f32" and a[0] == 8]
    b8 = [a for n, a in plan.bops if str(n) == "bbdm_winograd_gemm_f32" and a[0] == 8]
    w8 = [a for n, a in plan.bops if str(n) in ("bbdm_winograd_wgrad_finish_f32", "bbdm_winograd_wgrad_finish_bias_f32") and a[0] == 8]
    assert len(f8) >= 4 and len(b8) >= 4 and len(w8) == len(f8), (len(f8), len(b8), len(w8))
    m.winograd_train8 = 0           # (the m <= 6 plans of the same model: no launch on the tile; their gradients are the other tests')
    plan0 = m._plan_for(x, True)
    assert not any(a[0] == 8 for n, a in list(plan0.ops) + list(plan0.bops) if str(n).startswith("bbdm_winograd"))
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items()}
    (O.unet_forward(sdg, O.UNetSpec(**up), x, t, None) * dout).sum().backward()
    gmax = max(float(v.grad.abs().max()) for v in sdg.values())
    worst = 0.0
    for k, p in m.named_parameters():
        ref = sdg[k].grad
        e8 = float((p.grad - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * gmax)
        worst = max(worst, e8)
        if os.environ.get("BBDM_TEST_VERBOSE"):
            print(f"  {k}: {e8:.2e}")
        assert e8 < 1e-3, (k, e8)
    print(f"training plan on F(8x8): worst parameter gradient {worst:.2e} against the oracle's autograd")


def test_first_stage_plans_on_the_emulator():
    """The VQGAN first stage's encode / decode plans (bbdm_amd/first_stage_hip.py: the UNet plan's emitters driven by its own flag set)
    on the CPU-emulated kernels against the PyTorch first stage -- the build-container check that the first stage's flag set knows
    every switch the shared emitters read (a missing one surfaced only on the GPU box in round 3)."""
    import first_stage_cases as C
    C.encode_decode_parity(CPU, N=1, resolution=16, attn_resolutions=[8])


def test_sampling_loop_skips_input_copies_safely(monkeypatch):
    """tests/test_model_gpu.py's check of the x_next hand-over through the plan's input buffer, on the emulated kernels."""
    monkeypatch.setattr(M, "build", lambda rec, dev: _build(rec))
    M.test_sampling_loop_skips_input_copies_safely(CPU)


def test_sampling_under_inference_mode(monkeypatch):
    """tests/test_model_gpu.py's inference-mode sampling check (tensors without a version counter) on the emulated kernels."""
    monkeypatch.setattr(M, "build", lambda rec, dev: _build(rec))
    M.test_sampling_under_inference_mode(CPU)


@pytest.mark.parametrize("name", ["tiny_nocond", "tiny_xattn"])
def test_training_step_is_bitwise_reproducible(name):
    """Two training micro-steps from the same state give the same bits in the loss and every gradient (emulated kernels: the fibers of
    a block run in a fixed order, so this checks the plumbing of the limb accumulators -- buffer sizes, zeroing, the fold -- not the
    order-independence itself, which the -m gpu test of the same name covers)."""
    rec = load_case(name)
    m = _build(rec, train=True)
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else rec["y"]
    runs = []
    for rep in range(2):
        for p in m.parameters():
            p.grad = None
        loss, _ = m.p_losses(rec["x0"], rec["y"], ctx, rec["t"], rec["noise"])
        loss.backward()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    assert all(torch.equal(g, runs[1][1][k]) for k, g in runs[0][1].items())


def test_accumulation_in_place_matches_autograd(monkeypatch):
    """tests/test_training_gpu.py's check of the in-place gradient accumulation under dist_utils.accumulation_sync, emulated kernels."""
    real = T.build

    def build(rec, dev):
        m = real(rec, CPU)
        m.denoise_fn.hip_graph = False
        return m
    monkeypatch.setattr(T, "build", build)
    T.test_accumulation_in_place_matches_autograd(CPU, "tiny_nocond")


def test_long_sequence_attention_takes_the_presplit_form():
    """An AttentionBlock whose sequence the pre-split K / V form accepts (library option "attn_pipe" = 3: every T % 128 == 0; by default from
    T = 1024) is planned as bbdm_attention_kv_planes_f32 + bbdm_attention_planes_f32 on the idle Winograd scratch; the UNet output is
    bit-equal to the one-launch plan's, and the oracle's within the step tolerance."""
    import bbdm_amd
    from bbdm_amd import _lib
    import bbdm_oracle as O
    torch.manual_seed(7)
    params = dict(image_size=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(1,),
                  channel_mult=(1,), num_head_channels=32, use_scale_shift_norm=True, resblock_updown=True, condition_key="nocond")
    net = bbdm_amd.UNetModel(**params).eval()
    net.hip_graph = False
    x, t = torch.randn(2, 3, 16, 16), torch.tensor([3, 800])
    outs, names = [], []
    for mode, h2 in ((1, False), (3, False), (3, True)):
        with _lib.option("attn_pipe", mode), torch.no_grad():
            net._plans = {}
            net.attn_h2 = h2
            outs.append(net(x, timesteps=t).clone())
            names.append([str(getattr(n, "entry", n)) for n, _ in next(iter(net._plans.values())).ops
                          if "attention" in str(n) or "affine_bound" in str(n)])
    n_attn = len(names[0])                          # (input block, middle block, two output blocks)
    assert n_attn == 4 and names[0] == ["bbdm_attention_f32"] * n_attn
    assert names[1] == ["bbdm_attention_kv_planes_f32", "bbdm_attention_planes_f32"] * n_attn
    assert torch.equal(outs[0], outs[1])
    # the default: the pair on the fp16 planes under the projection's provable bound (GroupNorm bound x max row L1 of the qkv weight + max |bias|)
    assert names[2] == ["bbdm_h2_affine_bound_f32", "bbdm_attention_kv_planes_h2_f32", "bbdm_attention_planes_h2_f32"] * n_attn
    with torch.no_grad():
        ref = O.unet_forward({k: v.detach() for k, v in net.state_dict().items()}, O.UNetSpec(**params), x, t, None)
    assert parity_err(outs[1], ref) < M.STEP_TOL and parity_err(outs[2], ref) < M.STEP_TOL
    # the bound slots the attention launches read: set, finite, and -- loose by construction (sqrt(n_g) of the GroupNorm bound times the
    # L1 / typical ratio of a weight row) -- inside the 2^17 the pair absorbs above typical values of O(1)
    import ctypes
    plan = next(iter(net._plans.values()))
    slots = [args[-1] for n, args in plan.ops if str(getattr(n, "entry", n)) == "bbdm_attention_planes_h2_f32"]
    bounds = [ctypes.c_float.from_address(s.resolve()).value for s in slots]
    assert len(bounds) == n_attn and all(0.0 < b < 2.0 ** 17 for b in bounds), bounds


def test_forward_on_f8_tiles(monkeypatch):
    """UNetModel.winograd = 8: the inference forward of the large 3x3 layers on F(8x8, 3x3) (csrc/winograd_math.h: ten points; forced here
    onto a 16x16 model with 128-channel layers).  Within the step tolerance of the oracle and of the default plan; the training plan of
    the same model keeps the tiles that have a gradient side."""
    import bbdm_amd
    import bbdm_oracle as O
    from fixture_weights import synth_weights
    real = bbdm_amd.unet.winograd_tile

    def wt(N, H, W, cin, cout, max_m=6, small=True, allow8=False):
        if allow8 and max_m >= 8 and cin % 16 == 0 and cin >= 128 and cout % 128 == 0 and H >= 16:
            return 8
        return 4 if (max_m >= 4 and cin >= 64 and cout >= 64 and H % 4 == 0) else 0
    monkeypatch.setattr(bbdm_amd.unet, "winograd_tile", wt)
    up = dict(image_size=16, in_channels=4, model_channels=128, out_channels=4, num_res_blocks=1, attention_resolutions=(),
              channel_mult=(1, 1), conv_resample=True, dims=2, num_heads=2, num_head_channels=-1, use_scale_shift_norm=True,
              resblock_updown=True, use_spatial_transformer=False, context_dim=None, condition_key="nocond")
    m = bbdm_amd.unet.UNetModel(**up)
    sd = synth_weights([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 88)
    m.load_state_dict(sd, strict=True)
    m.hip_graph = False
    m.eval()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 4, 16, 16, generator=g)
    t = torch.tensor([7, 900])
    outs = {}
    for cap in (8, 6):
        m.winograd = cap
        with torch.no_grad():
            outs[cap] = m(x, timesteps=t, context=None).clone()
        plan = m._plan_for(x, False)
        tiles = sorted({a[0] for n, a in plan.ops if str(n) == "bbdm_winograd_gemm_f32"})
        assert (8 in tiles) == (cap == 8), tiles
    ref = O.unet_forward(sd, O.UNetSpec(**up), x, t, None)
    e8, e6 = parity_err(outs[8], ref), parity_err(outs[6], ref)
    print(f"tiny UNet on F(8x8,3x3): {e8:.2e} (default plan {e6:.2e})")
    assert e8 < 1e-3 and e6 < M.STEP_TOL
    m.winograd, m.winograd_train8 = 8, 0        # (training plans take the tile under winograd_train8: test_training_plan_on_f8_tiles)
    m.train()
    tplan = m._plan_for(x, True)
    assert 8 not in {a[0] for n, a in list(tplan.ops) + list(tplan.bops) if str(n) == "bbdm_winograd_gemm_f32"}
    assert real(16, 64, 64, 1024, 1024, 8, allow8=True) == 8 and real(16, 64, 64, 1024, 1024, 8) == 6 \
        and real(16, 64, 64, 1024, 1024, 6, allow8=True) == 6 and real(16, 256, 256, 128, 128, 8, allow8=True) == 8 \
        and real(32, 32, 32, 512, 512, 8, allow8=True) == 8 and real(32, 16, 16, 1024, 1024, 8, allow8=True) == 4
