"""Generate the golden fixtures under tests/golden/ by running the REAL reference (/root/reference).

TEST INFRASTRUCTURE.  Run in the build container only (the reference is not on the GPU box):

    python oracle/make_golden.py

The reference has no tests or golden vectors of its own, so the executable reference
module is the pin: this script imports it unmodified, drives ``BrownianBridgeModel`` with small
configurations that still exercise every code path of the hot path (ResBlock plain/up/down/1x1-skip,
AttentionBlock with legacy + new order, concat and nocond conditioning, all three objectives, both
loss types, both m_t schedules, skip/no-skip step tables) and stores inputs, weights and outputs.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]
from fixture_weights import synth_weights  # noqa: E402  (tests/fixture_weights.py)
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def d2n(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, d2n(v) if isinstance(v, dict) else v)
    return ns


def base_cfg():
    cfg = yaml.load(open(os.path.join(REF, "configs", "Template-BBDM.yaml")), Loader=yaml.FullLoader)
    return cfg["model"]


CASES = {
    # name: (UNetParams overrides, BB.params overrides, batch)
    "tiny_concat": (dict(image_size=16, in_channels=6, out_channels=3, model_channels=32,
                         channel_mult=(1, 2, 4), attention_resolutions=(4,), num_head_channels=32,
                         condition_key="SpatialRescaler"),
                    dict(objective="grad", loss_type="l1", mt_type="linear", skip_sample=True, sample_step=20), 2),
    "tiny_nocond": (dict(image_size=8, in_channels=8, out_channels=8, model_channels=32,
                         channel_mult=(1, 2, 2), attention_resolutions=(2, 4), num_head_channels=16,
                         condition_key="nocond", use_new_attention_order=True),
                    dict(objective="noise", loss_type="l2", mt_type="sin", skip_sample=False, num_timesteps=50), 3),
    "tiny_ysubx": (dict(image_size=8, in_channels=6, out_channels=3, model_channels=32,
                        channel_mult=(1, 2), attention_resolutions=(), num_heads=4, num_head_channels=-1,
                        condition_key="first_stage", use_scale_shift_norm=False, resblock_updown=False),
                   dict(objective="ysubx", loss_type="l1", mt_type="linear", skip_sample=True, sample_step=10,
                        eta=0.5, max_var=0.7), 2),
    # SpatialTransformer at every attention site (self-attention + cross-attention to the 8x8 condition image's 64 pixel
    # tokens: 64 and 16 queries against 64 keys) + GEGLU feed-forward; openaimodel.py:556-565, attention.py:153-263
    "tiny_xattn": (dict(image_size=8, in_channels=6, out_channels=3, model_channels=32, channel_mult=(1, 2),
                        attention_resolutions=(1, 2), num_head_channels=16, use_spatial_transformer=True,
                        transformer_depth=1, context_dim=3, condition_key="SpatialRescaler"),
                   dict(objective="grad", loss_type="l1", mt_type="linear", skip_sample=True, sample_step=10), 2),
}


def randomize(model, seed):
    """Fill the UNet with tests/fixture_weights.synth_weights (regenerable from the seed, so the
    fixtures need not carry the weights)."""
    shapes = [(k, tuple(v.shape)) for k, v in model.denoise_fn.state_dict().items()]
    sd = synth_weights(shapes, seed)
    model.denoise_fn.load_state_dict(sd, strict=True)
    return shapes


def main():
    sys.path.insert(0, REF)
    if "omegaconf" not in sys.modules:            # openaimodel.py:480 imports it only to unwrap a ListConfig context_dim
        try:
            import omegaconf  # noqa: F401
        except ImportError:
            import types
            oc, lc = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")
            lc.ListConfig = type("ListConfig", (list,), {})
            oc.listconfig = lc
            sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = oc, lc
    import model.BrownianBridge.BrownianBridgeModel as M
    from model.BrownianBridge.BrownianBridgeModel import BrownianBridgeModel

    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])                     # python oracle/make_golden.py [case ...]: regenerate only these
    for ci, (name, (unet_over, bb_over, batch)) in enumerate(CASES.items()):
        if only and name not in only:
            continue
        cfg = base_cfg()
        cfg["BB"]["params"].update(bb_over)
        cfg["BB"]["params"]["UNetParams"].update(unet_over)
        gen = torch.Generator().manual_seed(1234 + ci)
        torch.manual_seed(99 + ci)
        net = BrownianBridgeModel(d2n(cfg)).eval()
        wseed = 4321 + ci
        shapes = randomize(net, wseed)
        up = cfg["BB"]["params"]["UNetParams"]
        S, C = up["image_size"], up["out_channels"]
        x0 = torch.randn(batch, C, S, S, generator=gen).clamp(-1, 1)
        y = torch.randn(batch, C, S, S, generator=gen).clamp(-1, 1)
        T = net.num_timesteps
        t = torch.randint(0, T, (batch,), generator=gen)
        noise = torch.randn(batch, C, S, S, generator=gen)
        context = None if up["condition_key"] == "nocond" else y

        rec = {"unet_params": up, "bb_params": {k: v for k, v in cfg["BB"]["params"].items() if k != "UNetParams"},
               "unet_shapes": shapes, "weight_seed": wseed,
               "buffers": {k: v.clone() for k, v in net.state_dict().items() if not k.startswith("denoise_fn.")},
               "steps": net.steps.clone(), "x0": x0, "y": y, "t": t, "noise": noise}

        with torch.no_grad():
            # UNet alone
            rec["unet_out"] = net.denoise_fn(x0, timesteps=t, context=context)
            # q_sample / p_losses
            x_t, obj = net.q_sample(x0, y, t, noise)
            rec["q_x_t"], rec["q_objective"] = x_t, obj
            loss, log = net.p_losses(x0, y, context, t, noise)
            rec["loss"], rec["x0_recon"] = loss.clone(), log["x0_recon"].clone()
            # p_sample at a few loop indices, eps injected through the module-level torch.randn_like
            # (p_sample takes no noise argument: BrownianBridgeModel.py:197)
            n_steps = len(net.steps)
            idx = sorted(set([0, 1, n_steps // 2, n_steps - 2, n_steps - 1]))
            eps = torch.randn(batch, C, S, S, generator=gen)
            orig = M.torch.randn_like
            M.torch.randn_like = lambda ref: eps
            try:
                rec["p_idx"], rec["p_eps"] = idx, eps
                rec["p_x_t"] = x_t
                outs = []
                for clip in (False, True):
                    for i in idx:
                        a, b = net.p_sample(x_t.clone(), y, context, i, clip_denoised=clip)
                        outs.append((clip, i, a.clone(), b.clone()))
                rec["p_out"] = outs
                # short free-running loop with a fixed noise for every step
                rec["loop_out"] = net.p_sample_loop(y, context=None, clip_denoised=True).clone()
            finally:
                M.torch.randn_like = orig
        path = os.path.join(OUT, f"{name}.pt")
        torch.save(rec, path)
        nparam = sum(v.numel() for v in net.state_dict().values())
        print(f"{name}: {nparam/1e6:.2f} M state floats, steps={n_steps}, loss={float(loss):.6f} -> {path} "
              f"({os.path.getsize(path)/1e6:.1f} MB)")

    if only:
        return
    # schedule known-answer values for both schedules at T=1000
    kat = {}
    for mt in ("linear", "sin"):
        for skip in (True, False):
            cfg = base_cfg()
            cfg["BB"]["params"].update(dict(mt_type=mt, skip_sample=skip))
            cfg["BB"]["params"]["UNetParams"].update(dict(model_channels=32, image_size=8, channel_mult=(1,),
                                                         attention_resolutions=(), num_head_channels=32))
            net = BrownianBridgeModel(d2n(cfg))
            kat[(mt, skip)] = {"steps": net.steps.clone(),
                               **{k: getattr(net, k).clone() for k in
                                  ("m_t", "m_tminus", "variance_t", "variance_tminus", "variance_t_tminus",
                                   "posterior_variance_t")}}
    torch.save(kat, os.path.join(OUT, "schedule_kat.pt"))
    print("schedule_kat.pt written")


if __name__ == "__main__":
    main()
