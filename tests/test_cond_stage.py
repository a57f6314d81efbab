"""``bbdm_amd.cond_stage.SpatialRescaler`` (condition_key 'SpatialRescaler'): same outputs, state_dict and gradients as the
reference's class (model/BrownianBridge/base/modules/encoders/modules.py:106-134) when the checkout is mounted; its contract
without it."""
import importlib
import sys

import pytest
import torch
import torch.nn.functional as F

from bbdm_amd.cond_stage import SpatialRescaler

CASES = [dict(n_stages=1, method="bilinear", multiplier=0.5, in_channels=3, out_channels=None, bias=False),
         dict(n_stages=2, method="nearest", multiplier=0.5, in_channels=3, out_channels=8, bias=True),
         dict(n_stages=1, method="bicubic", multiplier=2, in_channels=4, out_channels=4, bias=False),
         dict(n_stages=0, method="area", multiplier=0.5, in_channels=3, out_channels=5, bias=False)]


@pytest.mark.parametrize("kw", CASES)
def test_contract(kw):
    torch.manual_seed(0)
    m = SpatialRescaler(**kw)
    x = torch.randn(2, kw["in_channels"], 16, 24, requires_grad=True)
    y = m.encode(x)
    want = x
    for _ in range(kw["n_stages"]):
        want = F.interpolate(want, scale_factor=kw["multiplier"], mode=kw["method"])
    if kw["out_channels"] is not None:
        assert sorted(m.state_dict()) == (["channel_mapper.bias", "channel_mapper.weight"] if kw["bias"] else ["channel_mapper.weight"])
        want = F.conv2d(want, m.channel_mapper.weight, m.channel_mapper.bias)
    else:
        assert not m.state_dict()
    assert torch.equal(y, want)
    y.sum().backward()
    assert x.grad is not None and all(p.grad is not None for p in m.parameters())
    with pytest.raises(AssertionError):
        SpatialRescaler(method="lanczos")


@pytest.mark.reference
@pytest.mark.parametrize("kw", CASES)
def test_matches_the_reference_class(kw):
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    try:
        ref_cls = importlib.import_module("model.BrownianBridge.base.modules.encoders.modules").SpatialRescaler
    except Exception as e:                                  # the module imports optional packages (transformers, kornia, clip ...)
        pytest.skip(f"reference encoders module not importable here: {e!r}")
    torch.manual_seed(1)
    ref = ref_cls(**kw)
    m = SpatialRescaler(**kw)
    m.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(2, kw["in_channels"], 16, 24)
    assert torch.equal(m(x), ref(x)) and torch.equal(m.encode(x), ref.encode(x))


def test_latent_model_builds_its_own_rescaler_and_trains_it():
    """LatentBrownianBridgeModel.py:33-50: condition_key 'SpatialRescaler' -> cond_stage_model = SpatialRescaler(**CondStageParams),
    and get_parameters() hands the optimizer the rescaler's parameters next to the UNet's.  Construction only (CPU)."""
    import argparse
    import bbdm_amd
    from fixtures import load_case

    def ns(c):
        o = argparse.Namespace()
        for k, v in c.items():
            setattr(o, k, ns(v) if isinstance(v, dict) else v)
        return o

    rec = load_case("tiny_concat")
    up = dict(rec["unet_params"], condition_key="SpatialRescaler")
    cfg = ns({"BB": {"params": dict(rec["bb_params"], UNetParams=up)}, "VQGAN": {"params": {"ckpt_path": None}},
              "CondStageParams": dict(n_stages=1, method="bilinear", multiplier=0.5, in_channels=3, out_channels=3, bias=True),
              "normalize_latent": False, "latent_before_quant_conv": False})
    m = bbdm_amd.LatentBrownianBridgeModel(cfg, vqgan=torch.nn.Conv2d(3, 3, 1))
    assert isinstance(m.cond_stage_model, bbdm_amd.SpatialRescaler)
    params = list(m.get_parameters())
    assert any(p is m.cond_stage_model.channel_mapper.weight for p in params)
    assert any(k.startswith("cond_stage_model.channel_mapper.") for k in m.state_dict())
