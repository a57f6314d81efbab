#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X: denoise-UNet sampling steps/s at 256x256 pixel-space BBDM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c5] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one ``BrownianBridgeModel.p_sample`` = one UNet forward on the local batch + the fused bridge update
(+ ``torch.randn_like`` for the step noise), exactly what the reference's sampling loop executes per iteration
(BrownianBridgeModel.py:171-201).  Inputs are synthetic (seed 1234, x,y = randn.clamp(-1,1); weights N(0,0.02) with
the zero-initialised modules randomised) and already resident in HBM when the timed region starts.

Multi-GPU: one process per GPU, image pairs shard naturally: every rank runs its own batch with no
data-path collective ("weak" scaling); the timed region is bracketed by barrier + synchronize on both sides and the
MAX over ranks is taken; value = steps all ranks completed / that time.

Rank 0 prints ONE JSON line with, besides the contract fields,
  roofline     : the dominant kernel -- the Winograd tile GEMMs on gemm_bf3p_pipe_kernel<NP = 2> (fp32-grade products on the 16-bit
                 matrix core: two fp16 planes per operand under a provable scale, THREE f16 MFMA terms per product, csrc/h2_split.h;
                 with UNetModel.gemm_h2 = False the six-term bf16x3 planes) -- fp32-equivalent FLOPs per launch / average launch
                 duration measured with HIP events on the launch stream, against the dense 16-bit MFMA peak / 3 (/ 6)
                 (MI355X_MICROARCH.md); `frac_step` = the whole step against the matrix peaks; `traffic` / `traffic_step` /
                 `mfma_util` (+ the scalars traffic_ratio_step, mfma_util_pct, dom_clock_GHz) from the committed rocprofv3 PMC
                 passes of this command (profiles/*_pmc_<workload>_*.json), shown only while those files carry the sha256 of the
                 library this process loaded -- else "stale";
  hip_graph    : the timed region replays the forward as one hipGraph (the product path); the per-launch events then come from an
                 eager pass of the same K steps right after it (`eager_profiled_ms_per_step`);
  f32mfma_ms_per_step : the same step with those GEMMs on the f32 MFMA (strict-fp32 A/B, 5 steps after the timed region);
  winograd6_ms_per_step : the same step on round 4's Winograd tiles (UNetModel.winograd = 6; parity["winograd6"] = its parity sample);
  bf16x3_ms_per_step : the same step on round 5's planes (UNetModel.gemm_h2 = False; parity["bf16x3"]);
                 c4: the training micro-step with F(8x8, 3x3) off in the training plan (UNetModel.winograd_train8 = 0; summary.c4_m6);
  parity       : image 0 of the benchmarked batch against the CPU path (c4: loss + named gradients against the oracle's autograd;
                 c3: + `loop`: worst single step and free-running drift over the first 24 steps of the sampling loop, default plan
                 and winograd = 6);
  cpu_baseline : the oracle (kind "port": oracle/bbdm_oracle.py, the validated restatement of the reference's CPU
                 path) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16)
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz

# BASELINE.json configs (UNetParams per configs/Template-*.yaml)
_UNET_PIXEL = dict(in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(32, 16, 8),
                   channel_mult=(1, 4, 8), conv_resample=True, dims=2, num_heads=8, num_head_channels=64,
                   use_scale_shift_norm=True, resblock_updown=True, use_spatial_transformer=False, context_dim=None,
                   condition_key="SpatialRescaler")
WORKLOADS = {
    # name: (description, UNetParams, latent/pixel channels, size, batch, skip_sample, sample_step)
    "c2": ("pixel-space BBDM 256x256, batch 16, 1000-step schedule (BASELINE.json configs[1])",
           dict(_UNET_PIXEL, image_size=256), 3, 256, 16, False, 1000),
    "c1": ("pixel-space BBDM 64x64, batch 4, 1000-step schedule (BASELINE.json configs[0])",
           dict(_UNET_PIXEL, image_size=64), 3, 64, 4, False, 1000),
    "c3": ("LBBDM-f4 latent 3x64x64, batch 32, 200 steps, UNet-only (BASELINE.json configs[2])",
           dict(_UNET_PIXEL, image_size=64, in_channels=3, condition_key="nocond"), 3, 64, 32, True, 200),
    "c5": ("LBBDM-f16 latent 8x16x16, batch 32, 200 steps, UNet-only (BASELINE.json configs[4], one GPU's shard)",
           dict(_UNET_PIXEL, image_size=16, in_channels=8, out_channels=8, attention_resolutions=(16, 8, 4),
                condition_key="nocond"), 8, 16, 32, True, 200),
}
WORKLOADS["c4"] = ("LBBDM-f4 training step (forward + backward + Adam), latent 3x64x64, batch 32 per GPU, DDP gradient "
                   "all-reduce when --gpus > 1 (BASELINE.json configs[3])",
                   dict(_UNET_PIXEL, image_size=64, in_channels=3, condition_key="nocond"), 3, 64, 32, True, 200)
BB = dict(mt_type="linear", objective="grad", loss_type="l1", sample_type="linear", num_timesteps=1000, eta=1.0,
          max_var=1.0)
# first stages of the latent workloads (configs/Template-LBBDM-f4.yaml:55-72, Template-LBBDM-f16.yaml:54-75)
FIRST_STAGE = {
    "c3": dict(embed_dim=3, n_embed=8192, ddconfig=dict(double_z=False, z_channels=3, resolution=256, in_channels=3, out_ch=3,
                                                         ch=128, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=[],
                                                         dropout=0.0)),
    "c5": dict(embed_dim=8, n_embed=16384, ddconfig=dict(double_z=False, z_channels=8, resolution=256, in_channels=3, out_ch=3,
                                                          ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2,
                                                          attn_resolutions=[16], dropout=0.0)),
}


def first_stage_pipeline(workload, batch, dev, ms_per_step, nsteps_table):
    """Whole LBBDM pipeline per batch (LatentBrownianBridgeModel.sample, LatentBrownianBridgeModel.py:103-132): encode the
    condition image, `nsteps_table` UNet steps (timed above), quantize + decode -- the first stage on the HIP kernels
    (bbdm_amd/first_stage_hip.py), random-init weights of the template's VQGAN geometry."""
    from bbdm_amd.first_stage_hip import VQModel
    torch.manual_seed(7)
    vq = VQModel(**FIRST_STAGE[workload]).eval().to(dev)
    x = torch.randn(batch, 3, 256, 256, device=dev).clamp(-1, 1)
    z = vq.encode_latent(x)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps
    enc, dec = timed(lambda: vq.encode_latent(x)), timed(lambda: vq.decode_latent(z))
    total_ms = enc + nsteps_table * ms_per_step + dec
    return {"encode_ms": enc, "decode_ms": dec, "unet_ms": nsteps_table * ms_per_step, "latent": list(z.shape[1:]),
            "imgs_per_sec_per_gpu": batch / (total_ms * 1e-3),
            "note": "encode(x_cond) + schedule_steps x the timed UNet step + quantize/decode, batch of 256x256 images"}


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


def synth_state(model_unet, seed=777):
    tests_dir = os.path.join(ROOT, "tests")                    # deterministic N(0, 0.02)-style weights: tests/fixture_weights.py
    if tests_dir not in sys.path:                              # (a weight generator shared with the fixtures, not the oracle)
        sys.path.insert(0, tests_dir)
    from fixture_weights import synth_weights
    shapes = [(k, tuple(v.shape)) for k, v in model_unet.state_dict().items()]
    return synth_weights(shapes, seed, w_std=0.02)


def make_inputs(batch, ch, size, seed=1234):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(batch, ch, size, size, generator=g).clamp(-1, 1)
    x_t = torch.randn(batch, ch, size, size, generator=g).clamp(-1, 1)
    return x_t, y


def _reference_model(up, skip, sstep, sd):
    """The reference's own BrownianBridgeModel (CPU) with our state_dict, when /root/reference is mounted (the build
    container); None on the GPU box.  Never part of the product path: bench.py's cpu_baseline leg only."""
    ref_root = os.environ.get("BBDM_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "model", "BrownianBridge")):
        return None
    try:
        if ref_root not in sys.path:
            sys.path.insert(0, ref_root)
        from model.BrownianBridge.BrownianBridgeModel import BrownianBridgeModel as RefModel
        m = RefModel(_ns({"BB": {"params": dict(BB, skip_sample=skip, sample_step=sstep, UNetParams=up)}}))
        m.denoise_fn.load_state_dict(sd, strict=True)
        return m.eval()
    except Exception as e:                      # pragma: no cover
        print(f"cpu_baseline: reference import failed ({e!r}); using the oracle port", file=sys.stderr)
        return None


class _CpuPath:
    """p_sample of the reference (kind 'reference') or of oracle/bbdm_oracle.py (kind 'port') on the host cores."""

    def __init__(self, up, skip, sstep, sd):
        odir = os.path.join(ROOT, "oracle")                     # the cpu_baseline leg is the only user of oracle/ in this file
        if odir not in sys.path:
            sys.path.insert(0, odir)
        import bbdm_oracle as O
        self.ref = _reference_model(up, skip, sstep, sd)
        self.kind = "reference" if self.ref is not None else "port"
        self.ora = None if self.ref is not None else O.OracleBBDM(
            {"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**up), skip_sample=skip, sample_step=sstep, **BB)

    @torch.no_grad()
    def p_sample(self, x_t, y, ctx, i, noise):
        if self.ref is None:
            return self.ora.p_sample(x_t, y, ctx, i, clip_denoised=False, noise=noise)
        import model.BrownianBridge.BrownianBridgeModel as RM       # inject the step noise (SURVEY.md §8c protocol)
        orig = RM.torch.randn_like
        RM.torch.randn_like = lambda t, **k: noise
        try:
            return self.ref.p_sample(x_t=x_t, y=y, context=ctx, i=i, clip_denoised=False)
        finally:
            RM.torch.randn_like = orig


def _time_cpu_steps(path, x_t, y, ctx, eps, n_timed, i0=1):
    ts = []
    for k in range(n_timed):
        t0 = time.perf_counter()
        path.p_sample(x_t, y, ctx, i0 + k, eps)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts


def _parity_record(g_a, g_b, a_ref, b_ref, i_par, kind):
    """Both metrics of SURVEY.md §8c on image 0 of the benchmarked batch (every `rel_err*` entry is held against `bar`)."""
    def rel(a, b):
        a, b = a.double().cpu(), b.double()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    def l2(a, b):
        a, b = a.double().cpu(), b.double()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    return {"rel_err_x_tminus": rel(g_a, a_ref[0]), "rel_err_x0_recon": rel(g_b, b_ref[0]),
            "rel_err_l2_x_tminus": l2(g_a, a_ref[0]), "rel_err_l2_x0_recon": l2(g_b, b_ref[0]), "bar": 1e-3,
            "metric": "max|gpu - cpu| / max|cpu| and ||gpu - cpu||_2 / ||cpu||_2 (rel_err_l2_*) on image 0 of the benchmarked batch, "
                      f"one p_sample step (schedule index {i_par}), same weights / inputs / step noise",
            "against": kind}


def cpu_baseline(workload, sd, budget_s=30.0, parity_inputs=None):
    """the reference's CPU path (imported from /root/reference when mounted -> kind "reference", else
    the validated restatement oracle/bbdm_oracle.py -> kind "port") timed on this box's host cores.

    Bounded sample of the workload: ONE image of the batch at the workload's resolution (a 256x256 batch-16 CPU step
    would need ~35 GB of attention scores), 1 warm-up + >= 3 timed p_sample steps (median), reported in the metric's
    unit (batch-B steps/s = per-image rate / B).  For c2 additionally C1 exactly as BASELINE.json names it (64x64,
    batch 4, 1 warm-up + 5 timed steps, median).  The warm-up step doubles as the parity sample: when
    ``parity_inputs`` = (x_t, y, i, eps, gpu_x_next_row0, gpu_x0_recon_row0) is given, its output for image 0 is compared
    with row 0 of the GPU's step on the same batch."""
    desc, up, ch, size, batch, skip, sstep = WORKLOADS[workload]
    path = _CpuPath(up, skip, sstep, sd)
    threads = torch.get_num_threads()
    parity = None
    if parity_inputs is not None:
        x_t, y, i_par, eps, g_a, g_b, alt = parity_inputs
        x_t, y, eps = x_t[:1].cpu(), y[:1].cpu(), eps[:1].cpu()
    else:
        x_t, y = make_inputs(1, ch, size)
        eps = torch.randn(x_t.shape, generator=torch.Generator().manual_seed(99))
        i_par = 0
    ctx = None if up["condition_key"] == "nocond" else y
    t0 = time.perf_counter()
    a_ref, b_ref = path.p_sample(x_t, y, ctx, i_par, eps)
    warm = time.perf_counter() - t0
    if parity_inputs is not None:
        parity = _parity_record(g_a, g_b, a_ref, b_ref, i_par, path.kind)
        for name, (h_a, h_b) in (alt or {}).items():      # the same sample on the A/B plans ("winograd6", "bf16x3")
            parity[name] = {k: v for k, v in _parity_record(h_a, h_b, a_ref, b_ref, i_par, path.kind).items() if k.startswith("rel_err")}
    n_timed = max(3, min(20, int((budget_s - warm) / max(warm, 1e-3))))
    med, ts = _time_cpu_steps(path, x_t, y, ctx, eps, n_timed)
    out = {"value": 1.0 / (med * batch), "unit": "steps/s", "cores": threads, "kind": path.kind,
           "host_cpus": os.cpu_count(),
           "sample": f"{path.kind} p_sample, image 0 of the {batch}-image batch at {size}x{size}, 1 warm-up + {n_timed} "
                     f"timed steps (median {med:.2f} s per image-step, min {ts[0]:.2f}, max {ts[-1]:.2f}), scaled by "
                     f"1/{batch} to batch-{batch} steps/s",
           "s_per_image_step": med}
    if workload == "c2":
        d1, up1, ch1, size1, batch1, skip1, sstep1 = WORKLOADS["c1"]
        p1 = _CpuPath(up1, skip1, sstep1, sd)                   # same 237 M weights (the UNet is resolution-agnostic)
        x1, y1 = make_inputs(batch1, ch1, size1)
        e1 = torch.randn(x1.shape, generator=torch.Generator().manual_seed(98))
        p1.p_sample(x1, y1, y1, 0, e1)
        # a 64x64 batch-4 step is small: all host threads oversubscribe it (round 3: 2.70 s on 128 threads of a 256-CPU box against
        # 1.5 s on 8 cores at survey time).  One step per candidate thread count, the 5 timed steps at the best one.
        sweep = {}
        for nt in sorted({t for t in (8, 16, 32, 64, 128, threads) if 1 <= t <= (os.cpu_count() or 1)}):
            torch.set_num_threads(nt)
            sweep[nt] = _time_cpu_steps(p1, x1, y1, y1, e1, 1)[0]
        best_nt = min(sweep, key=sweep.get)
        torch.set_num_threads(best_nt)
        med1, ts1 = _time_cpu_steps(p1, x1, y1, y1, e1, 5)
        torch.set_num_threads(threads)
        out["c1_exact"] = {"config": d1, "s_per_step_median": med1, "steps_per_s": 1.0 / med1,
                           "img_steps_per_s": batch1 / med1, "full_1000_step_sample_min": 1000 * med1 / 60.0,
                           "timed_steps": 5, "warmup": 1, "kind": p1.kind, "threads": best_nt,
                           "thread_sweep_s_per_step": {str(k): v for k, v in sweep.items()}}
    return out, parity


def parity_only(workload, sd, parity_inputs):
    """The parity sample without the timing leg: image 0 of the benchmarked batch, one p_sample step on the CPU path."""
    desc, up, ch, size, batch, skip, sstep = WORKLOADS[workload]
    path = _CpuPath(up, skip, sstep, sd)
    x_t, y, i_par, eps, g_a, g_b, alt = parity_inputs
    x_t, y, eps = x_t[:1].cpu(), y[:1].cpu(), eps[:1].cpu()
    ctx = None if up["condition_key"] == "nocond" else y
    a_ref, b_ref = path.p_sample(x_t, y, ctx, i_par, eps)
    parity = _parity_record(g_a, g_b, a_ref, b_ref, i_par, path.kind)
    for name, (h_a, h_b) in (alt or {}).items():
        parity[name] = {k: v for k, v in _parity_record(h_a, h_b, a_ref, b_ref, i_par, path.kind).items() if k.startswith("rel_err")}
    return parity


def loop_parity(model, workload, sd, y, dev, steps=24):
    """The first ``steps`` steps of the sampling LOOP on the benchmarked batch (round-5 verdict item 1b): image 0 of the batch against
    the CPU path's own trajectory from x_T = y with the same per-step noise -- `worst_step`: the largest single-step error when both
    start the step from the CPU trajectory; `drift`: the free-running difference after ``steps`` steps -- on the default plan and on
    UNetModel.winograd = 6 (max-norm and L2 form, the larger; the full 200 steps: tests/test_loop_parity_gpu.py)."""
    desc, up, ch, size, batch, skip, sstep = WORKLOADS[workload]
    path = _CpuPath(up, skip, sstep, sd)
    ctx_c = None if up["condition_key"] == "nocond" else y[:1].cpu()
    g = torch.Generator().manual_seed(2024)
    noises = [torch.randn(y.shape, generator=g) for _ in range(steps)]
    traj, x0s = [y[:1].cpu()], []
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))                     # (one small image: all host threads oversubscribe it)
    try:
        for i in range(steps):
            nxt, x0r = path.p_sample(traj[-1], y[:1].cpu(), ctx_c, i, noises[i][:1])
            traj.append(nxt)
            x0s.append(x0r)
    finally:
        torch.set_num_threads(threads)

    def err(a, b):
        a, b = a.double().cpu(), b.double()
        return max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30)))
    out = {"steps": steps, "against": path.kind}
    ctx = None if up["condition_key"] == "nocond" else y
    orig_rl = torch.randn_like
    cur = {}
    torch.randn_like = lambda t, **k: cur["eps"]
    try:
        for name, wino in (("default", None), ("winograd6", 6)):
            keep = model.denoise_fn.winograd
            if wino is not None:
                model.denoise_fn.winograd = wino
            try:
                worst, img = 0.0, y
                for i in range(steps):
                    cur["eps"] = noises[i].to(dev)
                    xin = torch.cat([traj[i].to(dev), y[1:]], 0)
                    a, b = model.p_sample(xin, y, ctx, i, clip_denoised=False)
                    worst = max(worst, err(a[:1], traj[i + 1]), err(b[:1], x0s[i]))       # (x_{t-1} and x0_recon of the step)
                    img, _ = model.p_sample(img, y, ctx, i, clip_denoised=False)
                    img = img.clone()
                torch.cuda.synchronize(dev)
                out[name] = {"worst_step": worst, "drift": err(img[:1], traj[-1])}
            finally:
                model.denoise_fn.winograd = keep
            for k in list(model.denoise_fn._plans)[1:]:
                del model.denoise_fn._plans[k]
    finally:
        torch.randn_like = orig_rl
    return out


def training_parity(model, sd, up, skip, sstep, x0, y, dev, all_grads=False, loss_type="l2"):
    """c4: loss and named parameter gradients of ONE micro-step on the benchmarked batch (batch 32, the benchmarked training plan)
    against autograd on the oracle (CPU, the whole batch).  l2 loss by default: d|t - p|/dp of the l1 loss is discontinuous, a single
    flipped sign of the 393 216 loss terms moves every gradient by ~1e-3 of its size; the loss kernel is the
    only thing that differs between the two.  ``loss_type`` = "l1" (the templates' loss; tests): the oracle is differentiated with the
    sign pattern of the HIP forward -- loss = sum(sign * (target - pred)) / count, the L1 loss with its non-differentiable sign() frozen,
    the number of elements whose sign differs between the two forwards is returned as ``sign_flips`` -- and the loss VALUE is compared
    unmodified."""
    odir = os.path.join(ROOT, "oracle")
    if odir not in sys.path:
        sys.path.insert(0, odir)
    import bbdm_oracle as O
    batch = x0.shape[0]
    g = torch.Generator().manual_seed(2468)
    t = torch.randint(0, 1000, (batch,), generator=g)
    noise = torch.randn(x0.shape, generator=g)
    names = ["input_blocks.0.0.weight", "input_blocks.4.0.in_layers.2.weight", "middle_block.1.qkv.weight",
             "middle_block.2.out_layers.3.weight", "output_blocks.5.0.skip_connection.weight", "time_embed.0.weight",
             "out.2.weight", "out.2.bias"]
    names = [n for n in names if n in sd]
    if all_grads:       # (tests/test_fullsize_parity_gpu.py: every parameter of the UNet at the benchmarked batch; ~2x the oracle time)
        names = [n for n, _ in model.denoise_fn.named_parameters()]
    keep = model.loss_type
    model.loss_type = loss_type
    seen = {}
    hook = model.denoise_fn.register_forward_hook(lambda mod, inp, out: seen.__setitem__("pred", out.detach())) if loss_type == "l1" else None
    try:
        model.zero_grad(set_to_none=True)
        loss, _ = model.p_losses(x0, y, None, t.to(dev), noise.to(dev))
        loss.backward()
        torch.cuda.synchronize(dev)
        got = {n: dict(model.denoise_fn.named_parameters())[n].grad.detach().cpu().clone() for n in names}
        loss_gpu = float(loss.detach())
        model.zero_grad(set_to_none=True)
        sign = None
        if loss_type == "l1":
            with torch.no_grad():
                _, target_gpu = model.q_sample(x0, y, t.to(dev), noise.to(dev))
            sign = torch.sign(target_gpu.cpu() - seen["pred"].float().cpu())
    finally:
        model.loss_type = keep
        if hook is not None:
            hook.remove()
    # (the timed micro-steps have stepped the optimizer: the oracle takes the model's CURRENT weights, not the initial ones)
    cur = {k: v.detach().cpu().clone() for k, v in model.denoise_fn.state_dict().items()}
    sdg = {"denoise_fn." + k: (v.requires_grad_() if k in names else v) for k, v in cur.items()}
    ora = O.OracleBBDM(sdg, O.UNetSpec(**up), skip_sample=skip, sample_step=sstep, **dict(BB, loss_type=loss_type))
    t0 = time.perf_counter()
    flips = None
    if sign is None:
        loss_ref, _ = ora.p_losses(x0.cpu(), y.cpu(), None, t, noise)
        loss_ref.backward()
    else:
        x_t, target = O.q_sample(ora.bufs, x0.cpu(), y.cpu(), t, noise, ora.objective)
        pred = ora.denoise(x_t, t, None)
        flips = int((sign != torch.sign(target - pred.detach())).sum())
        loss_ref = O.bb_loss(target, pred.detach(), "l1")                   # the loss VALUE: unmodified
        ((sign * (target - pred)).sum() / pred.numel()).backward()
    secs = time.perf_counter() - t0
    gmax = max(float(sdg["denoise_fn." + n].grad.abs().max()) for n in names)
    errs = {}
    for n in names:
        ref = sdg["denoise_fn." + n].grad
        errs[n] = float((got[n] - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * gmax)
    lr = float(loss_ref.detach())
    return {"rel_err_loss": abs(loss_gpu - lr) / max(abs(lr), 1e-30), "rel_err_grad_worst": max(errs.values()), "bar": 1e-3,
            "grad_errors": errs, "loss_gpu": loss_gpu, "loss_cpu": lr, "sign_flips": flips,
            "metric": f"one training micro-step on the benchmarked batch ({batch} x {tuple(x0.shape[1:])}, {loss_type} loss, fixed t / noise): "
                      "|loss_gpu - loss_cpu| / |loss_cpu| and, per named parameter, max|g_gpu - g_cpu| / max(|g_cpu|max, 1e-3 x the "
                      "largest gradient magnitude)",
            "against": "port (oracle autograd)", "cpu_seconds": secs}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # >= 20 steps after 3 warm-ups
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--accumulate", type=int, default=4, help="c4: accumulate_grad_batches (Template-LBBDM-f4.yaml:9)")
    ap.add_argument("--sync-every-micro-step", action="store_true", help="c4: all-reduce on every micro-step (reference)")
    ap.add_argument("--torch-adam", action="store_true", help="c4: torch.optim.Adam instead of the fused Adam+EMA pass")
    ap.add_argument("--no-pipeline", action="store_true", help="c3 / c5: skip the first-stage (VQGAN encode / decode) timing")
    ap.add_argument("--cpu-only", action="store_true", help="run only the cpu_baseline leg (no GPU; build container)")
    ap.add_argument("--dump-ops", default=None, help="write the per-launch table (name, shape, ms, TFLOP/s) here")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="A/B runs: set a library option (include/bbdm_hip.h \"options\", e.g. attn_pipe=0) for the whole run")
    ap.add_argument("--set", action="append", default=[], metavar="ATTR=INT",
                    help="A/B runs: set a planner attribute of UNetModel (e.g. fp32_v_max_cout=0) before the first plan is built")
    ap.add_argument("--no-f32mfma", action="store_true", help="c2: skip the strict-f32-MFMA A/B steps after the timed region")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity check of the benchmarked batch against the oracle")
    ap.add_argument("--no-op-profile", action="store_true",
                    help="skip the eager per-launch-event pass after the timed region (rocprofv3 traces of the graph replay alone)")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline run (c2, one GPU): do not time the other BASELINE.json configs (c1, c3, c5, c4) after it")
    return ap.parse_args(argv)


# the other BASELINE.json configs, timed after the headline in the default run: (workload, warm-up, timed steps)
EXTRA_WORKLOADS = (("c1", 5, 20), ("c3", 5, 20), ("c5", 5, 20), ("c4", 1, 8))


def main():
    args = parse_args()

    if args.cpu_only:                      # build-container check of the cpu_baseline leg (no GPU needed)
        import bbdm_amd
        desc, up, ch, size, batch, skip, sstep = WORKLOADS[args.workload]
        model = bbdm_amd.BrownianBridgeModel(_ns({"BB": {"params": dict(BB, skip_sample=skip, sample_step=sstep,
                                                                        UNetParams=up)}}))
        cb, _ = cpu_baseline(args.workload, synth_state(model.denoise_fn), args.cpu_budget)
        print(json.dumps({"cpu_baseline": cb}))
        return

    import bbdm_amd
    from bbdm_amd import dist_utils
    for kv in args.opt:
        from bbdm_amd import _lib as _bl
        k, v = kv.split("=")
        _bl.call("bbdm_set_option", k.encode(), int(v))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launches its own ranks (the reference spawns them itself too: main.py:100-104,
        # mp.spawn): re-exec under torch.distributed.run, one process per GPU, RCCL over xGMI, rendezvous on 127.0.0.1.
        sys.exit(dist_utils.self_launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank, local_rank, world = dist_utils.env_rank()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = dist_utils.init(backend="nccl")      # RCCL over xGMI; None when single-process
    env = (rank, local_rank, world, dev, dist)

    line = run_workload(args, env)
    # The default run (`python bench.py --gpus 1`) also times the other four BASELINE.json configs after the headline, so that every
    # configuration's step time, whole-step roofline fraction and parity sample are in the ONE line the driver records (they were
    # builder-run numbers in rounds 1-3).  Same code path as `--workload cN`; the cpu_baseline timing leg is the headline's only.
    if args.workload == "c2" and world == 1 and not args.no_extras and line is not None:
        line["workloads"] = {}
        for w, wu, st in EXTRA_WORKLOADS:
            torch.cuda.empty_cache()
            sub = argparse.Namespace(**vars(args))
            sub.workload, sub.warmup, sub.steps, sub.no_cpu, sub.dump_ops = w, wu, st, True, None
            t0 = time.perf_counter()
            r = run_workload(sub, env)
            line["workloads"][w] = {
                "config": r["config"]["workload"], "metric": r["metric"], "value": r["value"], "unit": r["unit"],
                "ms_per_step": r["ms_per_step"], "steps": st, "warmup": wu, "prime_steps": r["prime_steps"],
                "hip_graph": r["hip_graph"], "frac_step": r["roofline"]["frac_step"],
                "dominant_kernel_frac": r["roofline"]["frac"], "dominant_kernel_tflops": r["roofline"]["achieved"],
                "kernel_ms_per_step": r["kernel_ms_per_step"], "parity": r["parity"], "pipeline": r.get("pipeline"),
                "training": r.get("training"), "winograd6_ms_per_step": r.get("winograd6_ms_per_step"),
                "wall_s": time.perf_counter() - t0}
    if rank == 0 and line is not None:
        # LAST key of the line (the driver's stored record keeps the tail of stdout): every configuration's step time, whole-step
        # roofline fraction and worst parity error in <= 400 characters -- [ms_per_step, frac_step, worst rel_err]
        def _brief(r, frac):
            par = r.get("parity") or {}
            errs = [v for k, v in par.items() if k.startswith("rel_err")]
            return [round(r["ms_per_step"], 3), round(frac, 3), float(f"{max(errs):.2g}") if errs else None]
        line["summary"] = {args.workload: _brief(line, line["roofline"]["frac_step"])}
        for w, r in (line.get("workloads") or {}).items():
            line["summary"][w] = _brief(r, r["frac_step"])
        if line.get("winograd6_ms_per_step"):        # the headline on round 4's tiles: [ms_per_step, worst parity rel_err]
            w6 = (line.get("parity") or {}).get("winograd6") or {}
            line["summary"][args.workload + "_winograd6"] = [round(line["winograd6_ms_per_step"], 3),
                                                             float(f"{max(w6.values()):.2g}") if w6 else None]
        if line.get("bf16x3_ms_per_step"):           # the headline on round 5's six-term bf16x3 planes (UNetModel.gemm_h2 = False)
            b3 = (line.get("parity") or {}).get("bf16x3") or {}
            line["summary"][args.workload + "_bf16x3"] = [round(line["bf16x3_ms_per_step"], 3),
                                                          float(f"{max(b3.values()):.2g}") if b3 else None]
        c3r = line if args.workload == "c3" else (line.get("workloads") or {}).get("c3") or {}
        lp = ((c3r.get("parity") or {}).get("loop") or {})
        if lp.get("default"):                        # the first 24 steps of the C3 sampling loop: [worst single step, free-running drift]
            line["summary"]["c3_loop24"] = [float(f"{lp['default']['worst_step']:.2g}"), float(f"{lp['default']['drift']:.2g}")]
        c4r = line if args.workload == "c4" else (line.get("workloads") or {}).get("c4") or {}
        if c4r.get("winograd6_ms_per_step"):         # the training micro-step on the m <= 6 tiles (UNetModel.winograd_train8 = 0)
            line["summary"]["c4_m6"] = round(c4r["winograd6_ms_per_step"], 3)
        line["summary"]["units"] = "[ms_per_step, frac_step, worst parity rel_err]"
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_workload(args, env):
    """Time one workload on this rank's GPU; rank 0 returns the record (the JSON line of `--workload <name>`), other ranks None."""
    import bbdm_amd
    from bbdm_amd import dist_utils
    rank, local_rank, world, dev, dist = env
    desc, up, ch, size, batch, skip, sstep = WORKLOADS[args.workload]
    cfg = _ns({"BB": {"params": dict(BB, skip_sample=skip, sample_step=sstep, UNetParams=up)}})
    model = bbdm_amd.BrownianBridgeModel(cfg)
    sd = synth_state(model.denoise_fn)
    model.denoise_fn.load_state_dict(sd, strict=True)
    for kv in args.set:
        k, v = kv.split("=")
        if not hasattr(model.denoise_fn, k):
            raise SystemExit(f"bench.py --set: UNetModel has no attribute {k}")
        setattr(model.denoise_fn, k, type(getattr(model.denoise_fn, k))(int(v)) if getattr(model.denoise_fn, k) is not None else int(v))
    model = model.to(dev).eval()
    nparams = sum(p.numel() for p in model.denoise_fn.parameters())

    x_t, y = make_inputs(batch, ch, size, seed=1234 + rank)
    x_t, y = x_t.to(dev), y.to(dev)
    ctx = None if up["condition_key"] == "nocond" else y
    torch.manual_seed(1234 + rank)
    nsteps_table = len(model.steps)
    training = args.workload == "c4"

    if training:
        # the reference's training micro-step (runners/BaseRunner.py:398-423, BBDMRunner.py:164-176): net(x, x_cond) ->
        # loss.backward() -> every accumulate_grad_batches-th micro-step optimizer.step() + zero_grad(); EMA update every
        # update_ema_interval * accumulate_grad_batches micro-steps (Template-LBBDM-f4.yaml:9,43-47: 4, 8).  DDP
        # (BaseRunner.py:76) all-reduces the UNet gradients over RCCL/xGMI -- here only on the accumulation boundary
        # (dist_utils.accumulation_sync; --sync-every-micro-step restores the reference's behaviour for an A/B).
        from bbdm_amd.optim import EMA, FusedAdam
        model.train()
        net = model
        if dist is not None:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank)
        accumulate, ema_every = args.accumulate, 8
        if args.torch_adam:
            opt = torch.optim.Adam(model.get_parameters(), lr=1e-4, betas=(0.9, 0.999))
            ema = None
        else:
            opt = FusedAdam(model.get_parameters(), lr=1e-4, betas=(0.9, 0.999))
            ema = EMA(0.995)
            ema.register(model)

        sync_every = [bool(args.sync_every_micro_step)]      # (flipped for the A/B pass after the timed region when world > 1)
        opt_events = []                                       # HIP events around optimizer.step() + zero_grad() of the timed steps

        def step(i, img):
            gstep = i + 1                                     # the runner's 1-based global_step
            with dist_utils.accumulation_sync(net, gstep, 1 if sync_every[0] else accumulate):
                loss, _ = net(x_t, y)
                loss.backward()
            if gstep % accumulate == 0:
                eo = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                eo[0].record()
                if ema is not None and gstep % (ema_every * accumulate) == 0:
                    opt.step(ema=ema, ema_with_decay=False)   # before start_ema_step (30000): shadow = weights
                else:
                    opt.step()
                opt.zero_grad(set_to_none=True)
                eo[1].record()
                opt_events.append(eo)
            return loss.detach().reshape(1)
    else:
        def step(i, img):
            out, _ = model.p_sample(img, y, ctx, i % (nsteps_table - 1), clip_denoised=False)
            return out

    state = {"img": x_t}
    # c4: one accumulation cycle before the warm-up, so that the FIRST optimizer.step() -- which allocates and zero-fills the Adam
    # moments of all 560 parameter tensors and builds the fused pass's chunk table (~0.16 s, once per run) -- is not inside the K
    # timed steps (with the default W = 3 it fell on timed step 1 and added 8 ms to every step's average; measured round 3).
    # The K timed steps hold K / accumulate_grad_batches optimizer steps whatever the alignment.
    prime = args.accumulate if training else 0
    for i in range(prime + args.warmup):
        state["img"] = step(i, state["img"])
    prof = []
    # Per-launch HIP events are recorded INSIDE the timed region unless the forward is replayed as a hipGraph (small
    # latents in rounds 1-2, every inference plan since round 3) -- there the timed region is the plain product path and the events come from a second,
    # eager pass of the same K steps.
    plan_graph = (not training) and next(iter(model.denoise_fn._plans.values()))._want_graph()
    model.denoise_fn.op_profile = None if (training or plan_graph) else prof

    marks = []          # training: a HIP event after every timed micro-step (boundary vs non-boundary step time, see `training` below)

    def timed(collect=None):
        if collect is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            collect.append((0, e))
        for i in range(args.steps):
            state["img"] = step(prime + args.warmup + i, state["img"])
            if collect is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                collect.append((prime + args.warmup + i + 1, e))

    # barrier + synchronize on both sides, MAX over ranks (bbdm_amd/dist_utils.py)
    if training:
        opt_events.clear()
    elapsed = dist_utils.timed_region((lambda: timed(marks)) if training else timed, dist, dev)
    training_info = None
    if training:
        # Per-micro-step times from the events: the accumulation BOUNDARY micro-step carries the gradient all-reduce (DDP; the other
        # micro-steps run under no_sync) and the optimizer step; what the all-reduce adds beyond the backward it overlaps is
        # boundary - non-boundary - optimizer.  With --gpus N > 1 a second pass of the same K steps reduces on EVERY micro-step (the
        # reference's behaviour, runners/BaseRunner.py:412-417) for the A/B.
        durs = [(g, a.elapsed_time(b)) for (_, a), (g, b) in zip(marks[:-1], marks[1:])]
        bnd = [d for g, d in durs if g % args.accumulate == 0]
        non = [d for g, d in durs if g % args.accumulate != 0]
        opt_ms = [a.elapsed_time(b) for a, b in opt_events]
        mean = lambda v: (sum(v) / len(v)) if v else None
        training_info = {"accumulate_grad_batches": args.accumulate,
                         "gradient_sync": "every micro-step" if sync_every[0] else "accumulation boundary only (no_sync on the others)",
                         "boundary_micro_step_ms": mean(bnd), "non_boundary_micro_step_ms": mean(non),
                         "optimizer_step_ms": mean(opt_ms),
                         "allreduce_exposed_ms": (mean(bnd) - mean(non) - (mean(opt_ms) or 0.0)) if (bnd and non and world > 1) else None,
                         "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 else None}
        if world > 1 and not sync_every[0]:
            sync_every[0] = True
            for i in range(args.accumulate):
                state["img"] = step(prime + args.warmup + args.steps + i, state["img"])
            training_info["sync_every_micro_step_ms_per_step"] = dist_utils.timed_region(timed, dist, dev) * 1e3 / args.steps
            sync_every[0] = False
    eager_ms = None
    if (plan_graph or training) and not args.no_op_profile:
        # (training: the timed region is the plain product path as well; forward AND gradient-plan launches are timed here)
        model.denoise_fn.op_profile = prof
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        timed()
        eb.record()
        torch.cuda.synchronize(dev)
        eager_ms = ea.elapsed_time(eb) / args.steps
    model.denoise_fn.op_profile = None
    img = state["img"]
    if not bool(torch.isfinite(img).all()):
        raise RuntimeError("non-finite sample")
    # the same step with the tile GEMMs / 1x1 layers on the f32 MFMA (UNetModel.gemm_bf3 = False): a driver-timed number for a reader who
    # does not accept the bf16x3 emulation as fp32 arithmetic (5 steps after 2 warm-ups, outside the timed region)
    f32mfma_ms = None
    if args.workload == "c2" and not training and world == 1 and not args.no_f32mfma and model.denoise_fn.gemm_bf3:
        model.denoise_fn.gemm_bf3 = False
        try:
            for i in range(2):
                state["img"] = step(i, state["img"])
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5):
                state["img"] = step(2 + i, state["img"])
            e1.record()
            torch.cuda.synchronize(dev)
            f32mfma_ms = e0.elapsed_time(e1) / 5
        finally:
            model.denoise_fn.gemm_bf3 = True
        plan_keys = list(model.denoise_fn._plans)
        for k in plan_keys[1:]:                    # drop the A/B plan's buffers again
            del model.denoise_fn._plans[k]
        torch.cuda.empty_cache()

    # ... and on round 4's Winograd tiles (UNetModel.winograd = 6: F(6x6, 3x3) instead of F(8x8, 3x3) on the large layers -- 7x closer
    # to the reference, ~10 % slower): the same 2 + 5 steps; its parity sample is taken next to the headline's below
    winograd6_ms = None
    if args.workload in ("c2", "c3") and not training and world == 1 and not args.no_f32mfma and model.denoise_fn.winograd >= 8:
        model.denoise_fn.winograd = 6
        try:
            for i in range(2):
                state["img"] = step(i, state["img"])
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5):
                state["img"] = step(2 + i, state["img"])
            e1.record()
            torch.cuda.synchronize(dev)
            winograd6_ms = e0.elapsed_time(e1) / 5
        finally:
            model.denoise_fn.winograd = 8
        for k in list(model.denoise_fn._plans)[1:]:
            del model.denoise_fn._plans[k]
        torch.cuda.empty_cache()

    # ... and on round 5's planes (UNetModel.gemm_h2 = False: every tile GEMM on the six-term bf16x3 split): the same 2 + 5 steps
    bf16x3_ms = None
    if args.workload in ("c2", "c3") and not training and world == 1 and not args.no_f32mfma and model.denoise_fn.gemm_h2:
        model.denoise_fn.gemm_h2 = False
        try:
            for i in range(2):
                state["img"] = step(i, state["img"])
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5):
                state["img"] = step(2 + i, state["img"])
            e1.record()
            torch.cuda.synchronize(dev)
            bf16x3_ms = e0.elapsed_time(e1) / 5
        finally:
            model.denoise_fn.gemm_h2 = True
        for k in list(model.denoise_fn._plans)[1:]:
            del model.denoise_fn._plans[k]
        torch.cuda.empty_cache()

    # c4: the same micro-steps with the training plan on the tiles it had before F(8x8, 3x3) got its gradient side
    # (UNetModel.winograd_train8 = 0: m <= 6 forward, data gradient and weight gradient): one accumulation cycle to build and warm the
    # second plan, then two cycles timed, outside the timed region
    if training and world == 1 and not args.no_f32mfma and model.denoise_fn.winograd_train8 and model.denoise_fn.winograd >= 8:
        keep8 = model.denoise_fn.winograd_train8
        model.denoise_fn.winograd_train8 = 0
        try:
            base = prime + args.warmup + 2 * args.steps
            base += (-base) % args.accumulate                   # (start on an accumulation boundary)
            for i in range(args.accumulate):
                state["img"] = step(base + i, state["img"])
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(2 * args.accumulate):
                state["img"] = step(base + args.accumulate + i, state["img"])
            e1.record()
            torch.cuda.synchronize(dev)
            winograd6_ms = e0.elapsed_time(e1) / (2 * args.accumulate)
        finally:
            model.denoise_fn.winograd_train8 = keep8
        for k in list(model.denoise_fn._plans)[1:]:
            del model.denoise_fn._plans[k]
        torch.cuda.empty_cache()

    # ---- per-kernel accounting from the HIP events recorded inside the timed region -------------------------
    by = {}
    for name, e0, e1, fl in prof:
        d = by.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)            # ms
        d[2] += fl
    if args.dump_ops and rank == 0:
        plan = next(iter(model.denoise_fn._plans.values()))
        # (training: a profiled micro-step is the forward op list followed by the gradient plan's, bbdm_amd/unet.py: _backward_segment)
        ops_all = list(plan.ops) + (list(plan.bops) if training else [])
        nops = len(ops_all)
        agg = {}
        for j, (name, e0, e1, fl) in enumerate(prof):
            k = j % nops
            a = agg.setdefault(k, [name, 0.0, fl])
            a[1] += e0.elapsed_time(e1) / args.steps
        with open(args.dump_ops, "w") as f:
            f.write("| # | op | shape | ms | TFLOP/s |\n|---|---|---|---|---|\n")
            for k in sorted(agg):
                name, ms, fl = agg[k]
                oargs = ops_all[k][1]
                bwd = name.endswith(":bwd")
                name = name[:-4] if bwd else name
                if str(ops_all[k][0]) != name:
                    raise RuntimeError(f"--dump-ops: profile entry {k} is {name}, the plan's op is {ops_all[k][0]}")
                if name == "bbdm_conv1x1_bf3_f32":
                    ent = getattr(ops_all[k][0], "entry", name)
                    shp = "pixels{} {}->{} k1 ({})".format(*oargs[8:11], "fp16 pair" if "_h2" in ent else "bf16x3")
                elif name == "bbdm_conv2d_nhwc_f32":
                    shp = "N{} {}x{} {}->{} k{}".format(*oargs[15:21])
                elif name == "bbdm_winograd_gemm_f32":
                    from bbdm_amd.unet import wino_planes
                    shp = "N{} {}x{} {}->{} F({m}x{m},{r}x{r}) {p} GEMMs".format(*oargs[4:9], m=oargs[0], r=2 if oargs[0] == 7 else 3,
                                                                                 p=wino_planes(oargs[0]))
                elif name == "bbdm_winograd_input_f32":
                    shp = "F{} N{} {}x{} C{} up{} fusedGN{}".format(oargs[0], oargs[9], oargs[10], oargs[11], oargs[12],
                                                                    oargs[8], int(oargs[4] is not None))
                elif name == "bbdm_winograd_output_f32":
                    shp = "F{} N{} {}x{} C{}".format(oargs[0], *oargs[8:12])
                elif name == "bbdm_attention_f32":
                    shp = "N{} T{} heads{} ch{}".format(*oargs[5:9])
                elif name == "bbdm_groupnorm_apply_f32":
                    shp = "N{} {}x{} C{} silu{} rs{}".format(oargs[9], oargs[10], oargs[11], oargs[12], oargs[15], oargs[16])
                elif name == "bbdm_groupnorm_stats_f32":
                    shp = "N{} HW{} C{}".format(oargs[3], oargs[4], oargs[5])
                elif name == "bbdm_conv_wgrad_f32":
                    shp = "N{} {}x{} {}->{} k{}".format(*oargs[8:14])
                elif name == "bbdm_gemm_bf3p_tn_f32":
                    shp = "P{} tiles{} {}x{}".format(*oargs[3:7])
                elif name == "bbdm_groupnorm_bwd_f32":
                    shp = "N{} {}x{} C{} silu{} rs{}".format(oargs[19], oargs[20], oargs[21], oargs[22], oargs[25], oargs[26])
                else:
                    shp = ""
                tf = fl / (ms * 1e-3) / 1e12 if ms > 0 and fl else 0.0
                f.write(f"| {k} | {name.replace('bbdm_', '')}{':bwd' if bwd else ''} | {shp} | {ms:.3f} | {tf:.1f} |\n")
    # conv_igemm_f32 is launched by the direct convolutions and by the 16-GEMM stage of the Winograd layers; the
    # roofline counts the FLOPs the kernel EXECUTES (for a Winograd layer 4/9 of the direct-convolution FLOPs).
    def both(name):                 # forward launches + the same entry point launched by the gradient plan (":bwd")
        f, b = by.get(name, [0, 0.0, 0.0]), by.get(name + ":bwd", [0, 0.0, 0.0])
        return [f[0] + b[0], f[1] + b[1], f[2] + b[2]]
    direct = both("bbdm_conv2d_nhwc_f32")
    wino = both("bbdm_winograd_gemm_f32")
    conv = [direct[0] + wino[0], direct[1] + wino[1], direct[2] + wino[2]]
    executed_flops_per_step = sum(v[2] for v in by.values()) / max(1, args.steps)
    # SURVEY.md §8d's algorithmic figure is the direct-convolution count: a Winograd GEMM stands for 2.25x its FLOPs
    plan0 = next(iter(model.denoise_fn._plans.values()))
    def _direct_flops(oa):          # 2 * 9 * N * H * W * Cin * Cout of the layer a Winograd GEMM stands for
        cin = getattr(oa[2].t, "cin_true", oa[7])
        return 18.0 * oa[4] * oa[5] * oa[6] * cin * oa[8]
    extra = sum(_direct_flops(oa) - fl for (nm, oa), fl in zip(plan0.ops, plan0.op_flops)
                if nm == "bbdm_winograd_gemm_f32")              # per forward pass: direct count - executed count
    total_flops_per_step = executed_flops_per_step + (extra if not training else 0.0)
    # The dominant kernel.  With the default plan the Winograd tile GEMMs run on gemm_bf3_kernel (csrc/gemm_bf3.hip): fp32
    # arithmetic emulated EXACTLY-to-fp32-rounding on the BF16 matrix core (each operand split into 3 bf16, 6 product terms,
    # fp32 accumulate).  Its roofline is the dense bf16 MFMA peak; one fp32-equivalent FLOP costs 6 bf16 FLOP, so the bound
    # for the algorithmic (fp32) FLOPs is PEAK_BF16 / 6.  With gemm_bf3 = False (or where the shape gate rejects a layer)
    # the GEMMs run on conv_igemm_f32 (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s) together with the direct convolutions.
    all_ops = list(plan0.ops) + (list(plan0.bops) if training else [])
    entries = [getattr(nm, "entry", "") for nm, _ in all_ops if nm == "bbdm_winograd_gemm_f32"]
    is_h2 = lambda e: "_h2p_" in e
    h2_ops = sum(is_h2(e) for e in entries)
    bf3p_ops, bf3_ops = sum(e.endswith("bf3p_f32") or "bf3p_splitk" in e for e in entries), sum(e.endswith("bf3_f32") for e in entries)
    use_bf3 = h2_ops + bf3p_ops + bf3_ops > 0
    # fp32-equivalent FLOPs of the tile GEMMs by the instruction they issue (round 6): the fp16-pair planes cost THREE 16-bit MFMA
    # FLOPs per fp32-equivalent FLOP (csrc/h2_split.h), the bf16x3 planes six
    h2_share = (sum(fl for (nm, _), fl in zip(all_ops, list(plan0.op_flops) + [0.0] * (len(all_ops) - len(plan0.op_flops)))
                    if nm == "bbdm_winograd_gemm_f32" and is_h2(getattr(nm, "entry", ""))) /
                max(1.0, sum(fl for (nm, _), fl in zip(plan0.ops, plan0.op_flops) if nm == "bbdm_winograd_gemm_f32"))) if not training else 0.0
    terms = 3.0 if h2_ops > bf3p_ops + bf3_ops else 6.0
    if use_bf3 and terms == 3.0:
        dom, dom_name = wino, (f"gemm_bf3p_pipe_kernel<NP = 2> (v_mfma_f32_32x32x16_f16 x 3 terms on two fp16 planes per operand under a "
                               f"provable power-of-two scale = one fp32-grade product, csrc/h2_split.h; {h2_ops} launches per pass, "
                               f"{bf3p_ops + bf3_ops} on the bf16x3 planes)")
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    elif use_bf3:
        # the tile GEMMs run on csrc/gemm_bf3p.hip (both operands pre-split by their producers, LDS-DMA + MFMA main loop) where the
        # input transform writes the planes, else on csrc/gemm_bf3.hip (fp32 V split while staged): same arithmetic, bit for bit
        kname = ("gemm_bf3p_pipe_kernel" if bf3p_ops >= bf3_ops else "gemm_bf3_kernel")
        dom, dom_name = wino, (f"{kname} (v_mfma_f32_32x32x16_bf16 x 6 terms = one fp32-accurate product; {bf3p_ops} launches per "
                               f"pass on gemm_bf3p.hip, {bf3_ops} on gemm_bf3)")
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
    else:
        dom, dom_name, peak = conv, "conv_igemm_f32 (v_mfma_f32_32x32x2_f32)", PEAK_FP32_MFMA_TFLOPS
    conv_launches = dom[0]
    conv_ms = dom[1]
    flops_per_launch = dom[2] / max(1, conv_launches)
    avg_launch_ms = conv_ms / max(1, conv_launches)
    achieved = (flops_per_launch / (avg_launch_ms * 1e-3)) / 1e12 if avg_launch_ms > 0 else 0.0
    # whole step against the matrix peaks: time-at-peak of every MFMA kernel's work / step time
    c1x1 = both("bbdm_conv1x1_bf3_f32")                             # wide 1x1 convs / Linears on the same bf16x3 kernel
    # the attention FORWARD issues v_mfma_f32_32x32x16_bf16 as well (csrc/attention.hip, library option attn_bf3: 1 = Q K^T and P V -- head
    # widths 32 / 64 --, 2 or head width 16 = only Q K^T, 0 = f32 MFMA): its FLOPs are priced at the peak of the datatype it issues
    # (round 3 priced them at the f32 peak, which overstated frac_step by 4 points); the attention backward is on the f32 MFMA.
    attn = by.get("bbdm_attention_f32", [0, 0.0, 0.0])
    from bbdm_amd import _lib as _bl
    attn_mode = _bl.get_option("attn_bf3")
    attn_ch = next((oa[8] for nm, oa in plan0.ops if nm == "bbdm_attention_f32"), 64)
    attn_bf3_share = 0.0 if attn_mode == 0 else (1.0 if (attn_mode == 1 and attn_ch in (32, 64)) else 0.5)
    # executed FLOPs by the MFMA instruction that issues them, per profiled launch from the entry point the op is bound to (forward and
    # gradient plan alike): fp16-pair planes -> 2500 / 3, bf16x3 planes -> 2500 / 6, everything else -> the f32 MFMA peak
    cls = {"h2": 0.0, "bf3": 0.0}
    for pname, _e0, _e1, fl in prof:
        if not fl:
            continue
        base = str(pname)[:-4] if str(pname).endswith(":bwd") else str(pname)
        ent = getattr(pname, "entry", base)
        if base == "bbdm_attention_f32":
            if "_h2_" in ent:
                cls["h2"] += fl
            else:
                cls["bf3"] += attn_bf3_share * fl
        elif base in ("bbdm_winograd_gemm_f32", "bbdm_conv1x1_bf3_f32"):
            if "_h2" in ent:
                cls["h2"] += fl
            elif "bf3" in ent:
                cls["bf3"] += fl
    h2_flops, bf3_flops = cls["h2"] / max(1, args.steps), cls["bf3"] / max(1, args.steps)
    t_at_peak = (executed_flops_per_step - bf3_flops - h2_flops) / (PEAK_FP32_MFMA_TFLOPS * 1e12) + \
        bf3_flops / (PEAK_BF16_MFMA_TFLOPS / 6.0 * 1e12) + h2_flops / (PEAK_BF16_MFMA_TFLOPS / 3.0 * 1e12)
    # HBM-side traffic of the dominant kernel cannot be measured from inside the process: it comes from the committed
    # rocprofv3 PMC passes of this same command (profiles/*_pmc_<workload>_traffic.json), per launch, or null.
    traffic = None
    traffic_step = None
    import glob
    import hashlib
    from bbdm_amd import _lib as _bl0
    lib_sha = hashlib.sha256(open(_bl0.LIB_PATH, "rb").read()).hexdigest()
    stale = []                   # committed PMC files taken on ANOTHER build of the library: their fields are withheld
    # kernel-name prefix of the dominant kernel in the rocprofv3 tables (gemm_bf3_kernel also serves the few narrow 1x1 layers: only
    # counted where it IS the tile-GEMM kernel)
    dom_prefix = ("gemm_bf3p_" if bf3p_ops >= bf3_ops else "gemm_bf3_kernel") if use_bf3 else "conv_igemm_f32"
    try:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{args.workload}_traffic.json")))
        if cands:
            pm = json.load(open(cands[-1]))
            if pm.get("library_sha256") != lib_sha:
                stale.append(os.path.basename(cands[-1]))
                pm = {"kernels": {}}
            hits = [v for k, v in pm["kernels"].items() if k.startswith(dom_prefix)]
            if hits:                       # launch-weighted mean over the instantiations of the dominant kernel
                nl = sum(v["launches"] for v in hits)
                traffic = {"bytes_per_launch": sum(v["fabric_bytes_per_launch_corrected"] * v["launches"] for v in hits) / nl,
                           "source": os.path.basename(cands[-1]),
                           "note": "L2<->fabric bytes (FETCH_SIZE x2 + WRITE_SIZE), Infinity-Cache hits included"}
            tot = pm.get("totals")
            if tot:                        # whole step: every kernel's fabric bytes minus the one-time weight packing
                traffic_step = {"bytes_per_step": tot["bytes_per_step_excl_packing"], "source": os.path.basename(cands[-1]),
                                "algorithmic_bytes_per_step": tot.get("algorithmic_bytes_per_step"),
                                "note": "sum over all kernels of FETCH_SIZE x2 + WRITE_SIZE (L2<->fabric, Infinity-Cache hits "
                                        "included), one-time weight packing excluded; algorithmic = SURVEY.md 8(d)"}
    except Exception:
        traffic = None
    # MFMA utilisation of the dominant kernel from the committed PMC pass (tools/rocprof_counters.py --json)
    mfma_util = None
    try:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{args.workload}_mfma_util.json")))
        if cands:
            mu = json.load(open(cands[-1]))
            if mu.get("library_sha256") != lib_sha:
                stale.append(os.path.basename(cands[-1]))
                mu = {"kernels": {}}
            hits = [v for k, v in mu["kernels"].items() if k.startswith(dom_prefix)]
            hits = [v for v in hits if v.get("MfmaUtil%") is not None]
            if hits:
                wt = [v.get("launches", 0) * v.get("avg_us", 0.0) for v in hits]          # time each instantiation ran
                tot_w = sum(wt) or 1.0
                mfma_util = {"percent": sum(w * v["MfmaUtil%"] for w, v in zip(wt, hits)) / tot_w,
                             "effective_clock_GHz": sum(w * (v.get("clock_GHz") or 0.0) for w, v in zip(wt, hits)) / tot_w,
                             "weighting": "time-weighted over the kernel's instantiations",
                             "source": os.path.basename(cands[-1]),
                             "note": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): share of the kernel's cycles the "
                                     "matrix pipes are busy, at the clock the chip sustains under this load"}
    except Exception:
        mfma_util = None
    # algorithmic HBM bytes of the dominant kernel per launch: operands read once + result written once
    alg_bytes, alg_n = 0.0, 0
    for nm, oa in plan0.ops:
        if nm == "bbdm_winograd_gemm_f32":
            wm, gN, gH, gW, gci, gco = oa[0], *oa[4:9]
            from bbdm_amd.unet import wino_planes, wino_tiles
            P, T = wino_planes(wm), wino_tiles(wm, gN, gH, gW)
            ent = getattr(nm, "entry", "")
            a_bytes = 6.0 if (ent.endswith("bf3p_f32") or "bf3p_splitk" in ent) else 4.0     # V as three bf16 planes / two fp16 planes or fp32
            alg_bytes += P * T * (a_bytes * gci + 4.0 * gco) + (4.0 if (is_h2(ent) or not use_bf3) else 6.0) * P * gci * gco
            alg_n += 1
        elif nm == "bbdm_conv2d_nhwc_f32" and not use_bf3:
            gN, gH, gW, gci, gco, gks = oa[15:21]
            alg_bytes += 4.0 * (gN * gH * gW * (gci + gco) + gks * gks * gci * gco)
            alg_n += 1
    if traffic is not None and alg_n:
        traffic["algorithmic_bytes_per_launch"] = alg_bytes / alg_n
    ms_per_step = elapsed * 1e3 / args.steps
    devices = dist_utils.gather_device_info(dist, dev)          # per-rank device ids (+ RCCL version when world > 1)
    steps_per_s_job = dist_utils.aggregate_throughput(args.steps, elapsed, world)

    if rank == 0:
        line = {
            "metric": "denoise-UNet sampling steps/sec (one step = p_sample of the whole local batch: UNet forward + "
                      "Brownian-Bridge update) at 256x256 pixel-space BBDM" if args.workload == "c2" else
                      ("training micro-steps/sec (c4: forward + backward; fused Adam + EMA every accumulate_grad_batches-th)" if training else
                       f"denoise-UNet sampling steps/sec ({args.workload})"),
            "value": steps_per_s_job, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (tile GEMMs: fp32 operands as two fp16 planes under a provable power-of-two scale, 3 fp16-MFMA terms, fp32 "
                      "accumulate -- measured against fp64 MORE accurate than the six-term bf16x3 split and than the f32 MFMA; 1x1 layers "
                      "(skip / qkv / proj_out) and the long-sequence attention: the same pair; short-sequence, cross- and training "
                      "attention: bf16x3; all other kernels native fp32)") if (use_bf3 and terms == 3.0) else
                     ("f32 (tile GEMMs: fp32 operands split exactly into 3 bf16, 6 bf16-MFMA terms, fp32 accumulate -- fp32-accurate; "
                      "all other kernels native fp32)") if use_bf3 else "f32", "data": "synthetic (seed 1234 image pairs, random-init weights N(0,0.02))",
            "config": {"workload": desc, "batch_per_gpu": batch, "image_size": size, "unet_params_M": nparams / 1e6,
                       "schedule_steps": nsteps_table, "parallelism": f"dp{world} (independent image-pair shards)"},
            "devices": devices,
            "hip_graph": bool(plan_graph), "prime_steps": prime,
            # hip_graph: the timed region replays the forward as ONE hipGraph (the product path); the per-launch HIP events behind
            # `roofline` / `kernel_ms_per_step` then come from a second, eager pass of the same K steps right after it, whose own
            # step time (launch by launch, with two event records around every launch) is eager_profiled_ms_per_step
            "eager_profiled_ms_per_step": eager_ms,
            "steps_per_sec_per_gpu": args.steps / elapsed,
            "img_steps_per_sec": steps_per_s_job * batch,
            "imgs_per_sec_whole_job": steps_per_s_job * batch / nsteps_table,
            "tflops_algorithmic": total_flops_per_step / (ms_per_step * 1e-3) / 1e12,
            "tflops_executed": executed_flops_per_step / (ms_per_step * 1e-3) / 1e12,
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_note": (f"fp32-equivalent bound = dense 16-bit MFMA peak 2500 / {terms:.0f} ({terms:.0f} MFMA terms per "
                                       f"fp32-grade product); executed 16-bit rate = {terms:.0f} x achieved") if use_bf3 else
                                      "fp32-input MFMA peak (MI355X_MICROARCH.md)",
                         "executed_bf16_tflops": terms * achieved if use_bf3 else None,
                         "h2_share_of_tile_gemm_flops": h2_share if use_bf3 else None,
                         "frac_step": t_at_peak / (ms_per_step * 1e-3),
                         "frac_step_note": "whole step: time the MFMA work of every kernel would take at the matrix peak of the "
                                           "datatype it issues (2500 / 3 for launches on the fp16-pair planes -- tile GEMMs, 1x1 layers, "
                                           "attention, forward and gradient side --, 2500 / 6 for those on the bf16x3 planes; f32 MFMA for "
                                           "the rest) / step time",
                         "flops_per_step_by_mfma": {"fp16_pair": h2_flops, "bf16x3": bf3_flops,
                                                    "f32": executed_flops_per_step - bf3_flops - h2_flops},
                         "attention_bf3_share": attn_bf3_share,
                         "traffic": traffic, "traffic_step": traffic_step, "mfma_util": mfma_util,
                         # the same three as scalars (a record that flattens the line keeps them), or "stale" when the committed PMC
                         # pass was taken on another build of the library than the one this process loaded
                         "traffic_ratio_step": ((traffic_step["bytes_per_step"] / traffic_step["algorithmic_bytes_per_step"])
                                                if (traffic_step and traffic_step.get("algorithmic_bytes_per_step")) else
                                                ("stale" if stale else None)),
                         "mfma_util_pct": mfma_util["percent"] if mfma_util else ("stale" if stale else None),
                         "dom_clock_GHz": mfma_util["effective_clock_GHz"] if mfma_util else ("stale" if stale else None),
                         "pmc_library_sha256": lib_sha[:16], "pmc_stale_files": stale or None,
                         "launches_per_step": conv_launches / max(1, args.steps),
                         "gflop_per_launch": flops_per_launch / 1e9, "avg_launch_ms": avg_launch_ms,
                         "share_of_step_time": conv_ms / (elapsed * 1e3) if elapsed > 0 else None,
                         "flops_counted": "fp32-equivalent FLOPs executed: a Winograd layer's (m+2)^2 tile GEMMs = 4/9 (m=2), "
                                          "1/4 (m=4), 16/81 (m=6) or 25/144 (m=8) of its direct-convolution count",
                         "winograd_gemm_launches_per_step": wino[0] / max(1, args.steps),
                         "winograd_gemm_tflops": (wino[2] / (wino[1] * 1e-3) / 1e12) if wino[1] > 0 else None,
                         "direct_conv_tflops": (direct[2] / (direct[1] * 1e-3) / 1e12) if direct[1] > 0 else None,
                         "conv1x1_bf3_tflops": (c1x1[2] / (c1x1[1] * 1e-3) / 1e12) if c1x1[1] > 0 else None,
                         "direct_conv_peak": PEAK_FP32_MFMA_TFLOPS},
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in sorted(by.items())},
            "f32mfma_ms_per_step": f32mfma_ms,
            "winograd6_ms_per_step": winograd6_ms,
            "bf16x3_ms_per_step": bf16x3_ms,
            "training": training_info,
        }
        if args.workload in FIRST_STAGE and not args.no_pipeline:
            line["pipeline"] = first_stage_pipeline(args.workload, batch, dev, ms_per_step, nsteps_table)
        # parity sample (every workload, world 1, unless --no-parity): one more p_sample of the SAME batch with known step noise
        # (untimed), image 0 of which the CPU path recomputes (in its warm-up step when the cpu_baseline leg runs too); c4: the
        # loss and a handful of named gradients of the benchmarked batch against the oracle's autograd
        par_in = None
        if not training and not args.no_parity and world == 1:
            i_par = 431 % (nsteps_table - 1)
            eps = torch.randn(x_t.shape, generator=torch.Generator().manual_seed(4321)).to(dev)
            orig_rl = torch.randn_like
            torch.randn_like = lambda t, **k: eps
            try:
                g_a, g_b = model.p_sample(x_t, y, ctx, i_par, clip_denoised=False)
            finally:
                torch.randn_like = orig_rl
            torch.cuda.synchronize(dev)
            alt = {}
            for ab_name, ab_attr, ab_val, ab_on in (("winograd6", "winograd", 6, winograd6_ms is not None),
                                                    ("bf16x3", "gemm_h2", False, bf16x3_ms is not None)):
                if not ab_on:
                    continue
                keep = getattr(model.denoise_fn, ab_attr)          # the same sample on the A/B plan
                setattr(model.denoise_fn, ab_attr, ab_val)
                torch.randn_like = lambda t, **k: eps
                try:
                    h_a, h_b = model.p_sample(x_t, y, ctx, i_par, clip_denoised=False)
                finally:
                    torch.randn_like = orig_rl
                    setattr(model.denoise_fn, ab_attr, keep)
                torch.cuda.synchronize(dev)
                alt[ab_name] = (h_a[0].cpu(), h_b[0].cpu())
                for k in list(model.denoise_fn._plans)[1:]:
                    del model.denoise_fn._plans[k]
            par_in = (x_t, y, i_par, eps, g_a[0].cpu(), g_b[0].cpu(), alt)
        line["cpu_baseline"], line["parity"] = None, None
        if not args.no_cpu and world == 1:
            line["cpu_baseline"], line["parity"] = cpu_baseline(args.workload, sd, args.cpu_budget, par_in)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        elif par_in is not None:
            line["parity"] = parity_only(args.workload, sd, par_in)
        if args.workload == "c3" and line["parity"] is not None and not training and world == 1:
            line["parity"]["loop"] = loop_parity(model, args.workload, sd, y, dev)
        if training and not args.no_parity and world == 1:
            line["parity"] = training_parity(model, sd, up, skip, sstep, x_t, y, dev)
        par = line["parity"]
        if par is not None and not all(v < par["bar"] for k, v in par.items() if k.startswith("rel_err") and not isinstance(v, dict)):
            raise RuntimeError(f"bench parity check failed: {par}")
        return line
    return None


if __name__ == "__main__":
    main()
