"""LatentBrownianBridgeModel wrapper (GPU): API surface of LatentBrownianBridgeModel.py:19-147 with a small stand-in
first stage (the real VQGAN stays on PyTorch-ROCm and lives in the BBDM checkout; the wrapper only needs
``encoder / quant_conv / quantize / decode``).  The latent path must equal the oracle run on the same latents."""
import argparse

import pytest
import torch
import torch.nn as nn

from fixtures import few_threads, load_case, oracle_model, parity_err, rel_err

pytestmark = pytest.mark.gpu


class TinyFirstStage(nn.Module):
    """4x down / up conv autoencoder with a pass-through quantizer (same call surface as VQModel)."""

    def __init__(self, zc):
        super().__init__()
        self.encoder = nn.Sequential(nn.Conv2d(3, 16, 4, stride=4), nn.SiLU(), nn.Conv2d(16, zc, 1))
        self.quant_conv = nn.Conv2d(zc, zc, 1)
        self.post = nn.Sequential(nn.Conv2d(zc, 16, 1), nn.SiLU(), nn.ConvTranspose2d(16, 3, 4, stride=4))

    def quantize(self, z):
        return torch.round(z * 8) / 8, torch.zeros((), device=z.device), None

    def decode(self, z):
        return self.post(z)

    def forward(self, x):
        z, loss, _ = self.quantize(self.quant_conv(self.encoder(x)))
        return self.decode(z), loss


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


@pytest.mark.parametrize("normalize", [False, True])
def test_latent_wrapper_sample_and_train(normalize):
    import bbdm_amd
    dev = torch.device("cuda:0")
    rec = load_case("tiny_nocond")                       # latent UNet: 8 channels, 8x8, nocond
    cfg = _ns({"BB": {"params": dict(rec["bb_params"], UNetParams=rec["unet_params"])},
               "VQGAN": {"params": {"ckpt_path": None}}, "normalize_latent": normalize,
               "latent_before_quant_conv": False})
    torch.manual_seed(3)
    fs = TinyFirstStage(8)
    m = bbdm_amd.LatentBrownianBridgeModel(cfg, vqgan=fs)
    m.denoise_fn.load_state_dict({k[len("denoise_fn."):]: v for k, v in rec["state_dict"].items()
                                  if k.startswith("denoise_fn.")})
    m = m.to(dev)
    assert all(not p.requires_grad for p in m.vqgan.parameters())
    m.train()
    assert m.training and not m.vqgan.training           # disabled_train keeps the first stage in eval mode
    m.eval()
    assert [k for k in m.state_dict() if k.startswith("vqgan.")]
    assert len(list(m.get_parameters())) == len(list(m.denoise_fn.parameters()))
    if normalize:
        m.ori_latent_mean = torch.full((1, 8, 1, 1), 0.1, device=dev)
        m.ori_latent_std = torch.full((1, 8, 1, 1), 1.3, device=dev)
        m.cond_latent_mean = torch.full((1, 8, 1, 1), -0.2, device=dev)
        m.cond_latent_std = torch.full((1, 8, 1, 1), 0.9, device=dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 32, 32, generator=g).clamp(-1, 1).to(dev)
    x_cond = torch.randn(3, 3, 32, 32, generator=g).clamp(-1, 1).to(dev)
    z = m.encode(x_cond, cond=True)
    assert z.shape == (3, 8, 8, 8)
    # sampling: wrapper == decode(oracle loop on the same latents with the same per-step noise)
    eps = torch.randn(3, 8, 8, 8, generator=g)
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: eps.to(t.device)
    try:
        out = m.sample(x_cond, clip_denoised=False)
        mids, ones = m.sample(x_cond, clip_denoised=False, sample_mid_step=True)
    finally:
        torch.randn_like = orig
    ora = oracle_model(rec)
    with few_threads(8):                # (an 8x8 latent through a tiny UNet: all host threads only get in each other's way)
        lat = ora.p_sample_loop(z.cpu(), None, clip_denoised=False, noises=[eps] * len(ora.steps))
    want = m.decode(lat.to(dev), cond=False)
    assert out.shape == (3, 3, 32, 32) and parity_err(out.cpu(), want.cpu()) < 2e-3
    assert len(mids) == len(m.steps) + 1 and len(ones) == len(m.steps) and mids[-1].device.type == "cpu"
    # training step through the wrapper: loss is differentiable w.r.t. the UNet only
    m.train()
    loss, log = m(x, x_cond)
    loss.backward()
    assert all(p.grad is not None for p in m.denoise_fn.parameters())
    assert all(p.grad is None for p in m.vqgan.parameters())


def test_latent_with_builtin_first_stage():
    """No ``vqgan=``: the first stage is built from ``model_config.VQGAN.params`` by bbdm_amd.first_stage_hip.VQModel -- encode / decode
    on the HIP kernels (same schema as configs/Template-LBBDM-f4.yaml, shrunk)."""
    import bbdm_amd
    dev = torch.device("cuda:0")
    rec = load_case("tiny_nocond")
    dd = dict(double_z=False, z_channels=8, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=(1, 2, 2),
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    cfg = _ns({"BB": {"params": dict(rec["bb_params"], UNetParams=rec["unet_params"])},
               "VQGAN": {"params": {"ckpt_path": None, "embed_dim": 8, "n_embed": 128, "ddconfig": dd,
                                    "lossconfig": {"target": "torch.nn.Identity"}}},
               "normalize_latent": False, "latent_before_quant_conv": False})
    torch.manual_seed(4)
    m = bbdm_amd.LatentBrownianBridgeModel(cfg).to(dev)
    m.denoise_fn.load_state_dict({k[len("denoise_fn."):]: v for k, v in rec["state_dict"].items()
                                  if k.startswith("denoise_fn.")})
    assert any(k.startswith("vqgan.encoder.down.0.block.0.norm1") for k in m.state_dict())
    x_cond = torch.randn(2, 3, 32, 32).clamp(-1, 1).to(dev)
    assert m.encode(x_cond).shape == (2, 8, 8, 8)
    out = m.sample(x_cond, clip_denoised=False)
    assert out.shape == (2, 3, 32, 32) and bool(torch.isfinite(out).all())
    m.train()
    loss, _ = m(x_cond, x_cond.flip(0))
    loss.backward()
    assert all(p.grad is not None for p in m.get_parameters())
