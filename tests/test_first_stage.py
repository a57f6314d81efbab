"""bbdm_amd.first_stage.VQModel (PyTorch first stage) against the reference's VQModel on the CPU: identical state_dict
layout, identical encode / quantise / decode results under shared weights (build container only)."""
import argparse
import sys
import types

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.reference
REF = "/root/reference"


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


@pytest.mark.parametrize("attn,with_conv", [((16,), True), ((), True), ((8, 16), False)])
def test_first_stage_matches_reference(monkeypatch, attn, with_conv):
    import importlib.machinery
    pl = types.ModuleType("pytorch_lightning")
    pl.__spec__ = importlib.machinery.ModuleSpec("pytorch_lightning", None)
    pl.LightningModule = nn.Module
    monkeypatch.setitem(sys.modules, "pytorch_lightning", pl)
    monkeypatch.syspath_prepend(REF)
    from model.VQGAN.vqgan import VQModel as RefVQ
    from bbdm_amd.first_stage import VQModel
    dd = dict(double_z=False, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=(1, 2, 2),
              num_res_blocks=2, attn_resolutions=list(attn), dropout=0.0, resamp_with_conv=with_conv)
    params = dict(ckpt_path=None, embed_dim=4, n_embed=64, ddconfig=dd, lossconfig={"target": "torch.nn.Identity"})
    torch.manual_seed(0)
    ref = RefVQ(**vars(_ns(params))).eval()
    ours = VQModel(**vars(_ns(params))).eval()
    sd = ref.state_dict()
    assert {k: tuple(v.shape) for k, v in ours.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    g = torch.Generator().manual_seed(1)
    sd = {k: (torch.randn(v.shape, generator=g) * (0.3 if "embedding" in k else 0.05) + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
          for k, v in sd.items()}
    ref.load_state_dict(sd, strict=True)
    ours.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 32, 32, generator=g).clamp(-1, 1)
    with torch.no_grad():
        za, zb = ref.quant_conv(ref.encoder(x)), ours.quant_conv(ours.encoder(x))
        assert float((za - zb).abs().max()) < 1e-5 * float(za.abs().max())
        qa, la, ia = ref.quantize(za)
        qb, lb, ib = ours.quantize(za)
        assert torch.equal(ia[2].reshape(-1), ib[2].reshape(-1)) and torch.equal(qa, qb)
        assert abs(float(la) - float(lb)) < 1e-7
        da, db = ref.decode(qa), ours.decode(qa)
        assert float((da - db).abs().max()) < 2e-5 * float(da.abs().max())
        ra, rb = ref(x), ours(x)
        assert float((ra[0] - rb[0]).abs().max()) < 5e-4 * float(ra[0].abs().max())
