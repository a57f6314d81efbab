"""Sampling-loop parity at the real UNet size: the 237 M-parameter LBBDM-f4 UNet, 200-step
schedule, latent 3x64x64.

* per-step: the oracle's own trajectory x_t (CPU) is fed to BOTH implementations at sampled loop indices and the
  outputs of that single step are compared (bar: 1e-3 relative, BASELINE.json north_star) -- no accumulated drift;
* free-running: both loops run all 200 steps from the same y with the same per-step noise; the end-to-end drift is
  reported and bounded.

Both at batch 1 (small-problem plan: F(6x6) / F(4x4) tiles) and -- round 6 -- at BASELINE.json configs[2]'s own batch of 32, whose plan
runs the 64^2 and 32^2 levels on the default F(8x8, 3x3) tiles (>= 512 tiles per layer): image 0 of the batch carries the oracle's y and
noise (GroupNorm is per image, so one CPU trajectory checks the batch-32 plan), for ``winograd`` = 8 and 6.
"""
import argparse

import pytest
import torch

import bbdm_oracle as O
from fixture_weights import synth_weights
from fixtures import few_threads, parity_err, rel_err

pytestmark = pytest.mark.gpu

UP = dict(image_size=64, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
          attention_resolutions=(32, 16, 8), channel_mult=(1, 4, 8), conv_resample=True, dims=2, num_heads=8,
          num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True, use_spatial_transformer=False,
          context_dim=None, condition_key="nocond")
BB = dict(mt_type="linear", objective="grad", loss_type="l1", skip_sample=True, sample_type="linear", sample_step=200,
          num_timesteps=1000, eta=1.0, max_var=1.0)


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


def test_200_step_loop_per_step_and_free_running():
    import bbdm_amd
    dev = torch.device("cuda:0")
    m = bbdm_amd.BrownianBridgeModel(_ns({"BB": {"params": dict(BB, UNetParams=UP)}}))
    shapes = [(k, tuple(v.shape)) for k, v in m.denoise_fn.state_dict().items()]
    sd = synth_weights(shapes, 2024, w_std=0.02)
    m.denoise_fn.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    ora = O.OracleBBDM({"denoise_fn." + k: v for k, v in sd.items()}, O.UNetSpec(**UP), **BB)
    g = torch.Generator().manual_seed(7)
    y = torch.randn(1, 3, 64, 64, generator=g).clamp(-1, 1)
    n = len(ora.steps)
    assert n == 200 and len(m.steps) == 200
    noises = [torch.randn(1, 3, 64, 64, generator=g) for _ in range(n)]

    # oracle trajectory (CPU), keeping every x_t and x0_recon.  One 64x64 image per step is a small problem: the GPU box's 128
    # default threads oversubscribe it (0.93 s per step in round 3, 212 s for this test); 32 threads are faster.
    traj, x0s = [y], []
    with few_threads(32):
        for i in range(n):
            nxt, x0r = ora.p_sample(traj[-1], y, None, i, clip_denoised=True, noise=noises[i])
            traj.append(nxt)
            x0s.append(x0r)

    cur = {"eps": None}
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: cur["eps"].to(t.device)
    results = {}
    try:
        for batch, wino in ((1, 8), (32, 8), (32, 6)):
            m.denoise_fn.winograd = wino
            m.denoise_fn._plans = {}
            gb = torch.Generator().manual_seed(100 + batch)
            pad = lambda t: torch.cat([t, torch.randn(batch - 1, 3, 64, 64, generator=gb).clamp(-1, 1)], 0) if batch > 1 else t
            ys = pad(y).to(dev)
            fill = [torch.randn(batch - 1, 3, 64, 64, generator=gb) for _ in range(n)] if batch > 1 else None
            eps_of = lambda i: noises[i] if batch == 1 else torch.cat([noises[i], fill[i]], 0)
            # per-step parity on the oracle's trajectory (image 0; the other images start from their own y)
            worst = 0.0
            for i in list(range(0, n, 8)) + [n - 2, n - 1]:
                cur["eps"] = eps_of(i)
                xin = traj[i] if batch == 1 else torch.cat([traj[i], ys[1:].cpu()], 0)
                a, b = m.p_sample(xin.to(dev), ys, None, i, clip_denoised=True)
                a_ref, b_ref = traj[i + 1], x0s[i]             # (the oracle's own step from traj[i] with noises[i])
                worst = max(worst, parity_err(a[:1].cpu(), a_ref), parity_err(b[:1].cpu(), b_ref))
            plan = next(iter(m.denoise_fn._plans.values()))
            tiles = sorted({args[0] for name, args in plan.ops if str(name) == "bbdm_winograd_gemm_f32"})
            n8 = sum(1 for name, args in plan.ops if str(name) == "bbdm_winograd_gemm_f32" and args[0] == 8)
            # batch 32 on the default setting: the plan that is benchmarked -- most of its Winograd layers on F(8x8, 3x3)
            assert (n8 >= 20) == (batch == 32 and wino == 8), (batch, wino, tiles, n8)
            # free-running loop on the GPU
            img = ys
            for i in range(n):
                cur["eps"] = eps_of(i)
                img, _ = m.p_sample(img, ys, None, i, clip_denoised=True)
            drift = parity_err(img[:1].cpu(), traj[-1])
            results[(batch, wino)] = (worst, drift)
            print(f"200-step loop, batch {batch}, winograd <= {wino} (tiles {tiles}, {n8} layers on F(8x8)): worst per-step rel err {worst:.2e}; "
                  f"free-running end-to-end drift {drift:.2e}")
    finally:
        torch.randn_like = orig
    for (batch, wino), (worst, drift) in results.items():
        assert worst < 1e-3, (batch, wino, worst)
        assert drift < 1e-2, (batch, wino, drift)
    # the default tile must not cost the loop more than 5e-4 per step / 5e-3 end to end (round-5 verdict: its margin was unmeasured)
    assert results[(32, 8)][0] < 5e-4 and results[(32, 8)][1] < 5e-3, results[(32, 8)]
