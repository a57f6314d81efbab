"""Frozen VQGAN first stage for ``LatentBrownianBridgeModel`` -- plain PyTorch-ROCm, by design.

BASELINE.json's north_star leaves "the VQGAN encode/decode on PyTorch-ROCm" (2 encodes + 1 decode per 200 UNet
calls; SURVEY.md §2.1 row 6, §8f lists moving it onto the HIP kernels as the first "next" item).  Used as a drop-in
inside the BBDM checkout, ``LatentBrownianBridgeModel`` instantiates the checkout's own ``model.VQGAN.vqgan.VQModel``;
this module is the stand-alone equivalent so that ``bbdm_amd`` can run LBBDM without the checkout: same constructor
keywords (``configs/Template-LBBDM-*.yaml: model.VQGAN.params``), same ``state_dict`` keys / shapes (taming VQGAN
checkpoints load with ``strict=True``), same ``encoder / quant_conv / quantize / post_quant_conv / decoder / decode``
call surface.  Reference: model/VQGAN/vqgan.py:31-93, model/VQGAN/model.py:34-192,342-537, quantize.py:213-329.

Inference only: the first stage is frozen in BBDM (LatentBrownianBridgeModel.py:23-27; ``lossconfig`` is
``torch.nn.Identity`` in every template), so no GAN / perceptual losses are provided.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _cfg(c) -> dict:
    return dict(c) if isinstance(c, dict) else dict(vars(c))


def _norm(ch):
    return nn.GroupNorm(32, ch, eps=1e-6, affine=True)        # model.py:34-35


class ResnetBlock(nn.Module):
    """GN -> swish -> conv3 -> GN -> swish -> conv3, 1x1 ("nin") shortcut when the width changes (model.py:78-137)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.norm1, self.conv1 = _norm(cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = _norm(cout), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + h


class AttnBlock(nn.Module):
    """Single-head spatial self-attention with 1x1 q/k/v/proj convs (model.py:140-192)."""

    def __init__(self, ch):
        super().__init__()
        self.norm = _norm(ch)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(ch, ch, 1) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.norm(x)
        q, k, v = (m(y).reshape(b, c, h * w).transpose(1, 2) for m in (self.q, self.k, self.v))     # [b, hw, c]
        a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], scale=float(c) ** -0.5)[:, 0]
        return x + self.proj_out(a.transpose(1, 2).reshape(b, c, h, w))


class Downsample(nn.Module):
    """Stride-2 3x3 conv on a (0,1,0,1)-padded input, or 2x2 average pooling (model.py:56-75)."""

    def __init__(self, ch, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1))) if self.with_conv else F.avg_pool2d(x, 2, 2)


class Upsample(nn.Module):
    """Nearest x2 then an optional 3x3 conv (model.py:38-53)."""

    def __init__(self, ch, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


def _level(blocks, attns, resample_name=None, resample=None):
    m = nn.Module()
    m.block, m.attn = nn.ModuleList(blocks), nn.ModuleList(attns)
    if resample is not None:
        setattr(m, resample_name, resample)
    return m


def _middle(ch):
    m = nn.Module()
    m.block_1, m.attn_1, m.block_2 = ResnetBlock(ch, ch), AttnBlock(ch), ResnetBlock(ch, ch)
    return m


def _run_level(level, h):
    for i, blk in enumerate(level.block):
        h = blk(h)
        if len(level.attn):
            h = level.attn[i](h)
    return h


class Encoder(nn.Module):
    """model.py:342-433."""

    def __init__(self, *, ch, out_ch=None, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **_):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        widths = [ch * m for m in ch_mult]
        self.down = nn.ModuleList()
        cur, res = ch, resolution
        for lvl, wdt in enumerate(widths):
            blocks, attns = [], []
            for _ in range(num_res_blocks):
                blocks.append(ResnetBlock(cur, wdt))
                cur = wdt
                if res in attn_resolutions:
                    attns.append(AttnBlock(cur))
            last = lvl == len(widths) - 1
            self.down.append(_level(blocks, attns, "downsample", None if last else Downsample(cur, resamp_with_conv)))
            if not last:
                res //= 2
        self.mid = _middle(cur)
        self.norm_out = _norm(cur)
        self.conv_out = nn.Conv2d(cur, 2 * z_channels if double_z else z_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for level in self.down:
            h = _run_level(level, h)
            if hasattr(level, "downsample"):
                h = level.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(F.silu(self.norm_out(h)))


class Decoder(nn.Module):
    """model.py:436-537 (``up`` is stored lowest-resolution-last, i.e. indexed by level, like the reference)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels=None, resolution, z_channels, give_pre_end=False, **_):
        super().__init__()
        self.give_pre_end = give_pre_end
        widths = [ch * m for m in ch_mult]
        cur = widths[-1]
        res = resolution // 2 ** (len(widths) - 1)
        self.conv_in = nn.Conv2d(z_channels, cur, 3, padding=1)
        self.mid = _middle(cur)
        levels = []
        for lvl in reversed(range(len(widths))):
            blocks, attns = [], []
            for _ in range(num_res_blocks + 1):
                blocks.append(ResnetBlock(cur, widths[lvl]))
                cur = widths[lvl]
                if res in attn_resolutions:
                    attns.append(AttnBlock(cur))
            levels.insert(0, _level(blocks, attns, "upsample", Upsample(cur, resamp_with_conv) if lvl != 0 else None))
            if lvl != 0:
                res *= 2
        self.up = nn.ModuleList(levels)
        self.norm_out = _norm(cur)
        self.conv_out = nn.Conv2d(cur, out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for level in reversed(self.up):
            h = _run_level(level, h)
            if hasattr(level, "upsample"):
                h = level.upsample(h)
        if self.give_pre_end:
            return h
        return self.conv_out(F.silu(self.norm_out(h)))


class VectorQuantizer(nn.Module):
    """Nearest-codebook-entry quantiser (quantize.py:213-329, ``VectorQuantizer2`` without index remapping)."""

    def __init__(self, n_e, e_dim, beta=0.25, sane_index_shape=False, legacy=True):
        super().__init__()
        self.n_e, self.e_dim, self.beta, self.legacy, self.sane_index_shape = n_e, e_dim, beta, legacy, sane_index_shape
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def forward(self, z, temp=None, rescale_logits=False, return_logits=False):
        zl = z.permute(0, 2, 3, 1).contiguous()                       # b h w c
        flat = zl.view(-1, self.e_dim)
        e = self.embedding.weight
        # |z|^2 + |e|^2 - 2 z.e, evaluated in the reference's order so that ties / near-ties pick the same code
        d = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(e ** 2, dim=1) - 2 * (flat @ e.t())
        idx = torch.argmin(d, dim=1)
        zq = self.embedding(idx).view(zl.shape)
        if self.legacy:
            loss = torch.mean((zq.detach() - zl) ** 2) + self.beta * torch.mean((zq - zl.detach()) ** 2)
        else:
            loss = self.beta * torch.mean((zq.detach() - zl) ** 2) + torch.mean((zq - zl.detach()) ** 2)
        zq = (zl + (zq - zl).detach()).permute(0, 3, 1, 2).contiguous()
        if self.sane_index_shape:
            idx = idx.reshape(zq.shape[0], zq.shape[2], zq.shape[3])
        return zq, loss, (None, None, idx)

    def get_codebook_entry(self, indices, shape):
        zq = self.embedding(indices)
        return zq.view(shape).permute(0, 3, 1, 2).contiguous() if shape is not None else zq


class VQModel(nn.Module):
    """model/VQGAN/vqgan.py:31-93 (the parts BBDM uses)."""

    def __init__(self, ddconfig, lossconfig=None, n_embed=None, embed_dim=None, ckpt_path=None, ignore_keys=(),
                 image_key="image", colorize_nlabels=None, monitor=None, remap=None, sane_index_shape=False):
        super().__init__()
        if remap is not None:
            raise NotImplementedError("bbdm_amd.first_stage: codebook index remapping is not supported")
        dd = _cfg(ddconfig)
        self.image_key = image_key
        self.encoder = Encoder(**dd)
        self.decoder = Decoder(**dd)
        self.loss = nn.Identity()
        self.quantize = VectorQuantizer(n_embed, embed_dim, beta=0.25, sane_index_shape=sane_index_shape)
        self.quant_conv = nn.Conv2d(dd["z_channels"], embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, dd["z_channels"], 1)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        # GAN / perceptual-loss weights of a VQGAN training checkpoint are not part of the frozen first stage
        sd = {k: v for k, v in sd.items() if not k.startswith("loss.")}
        # like the reference (vqgan.py:66-75: strict=False): extra keys of a training checkpoint (model_ema.*, a colorize buffer)
        # are ignored with a note -- but a MISSING key would leave part of the first stage at its random initialisation and sample
        # garbage silently, so that raises
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if missing:
            raise RuntimeError(f"{path}: first-stage weights missing from the checkpoint: {sorted(missing)[:8]} ...")
        if unexpected:
            print(f"{path}: ignored {len(unexpected)} checkpoint entries that are not part of the first stage "
                  f"({sorted(unexpected)[:4]} ...)")
        print(f"Restored from {path}")

    def encode(self, x):
        return self.quantize(self.quant_conv(self.encoder(x)))

    def decode(self, quant):
        return self.decoder(self.post_quant_conv(quant))

    def decode_code(self, code_b):
        return self.decode(self.quantize.embedding(code_b).permute(0, 3, 1, 2).contiguous())

    def forward(self, x):
        quant, diff, _ = self.encode(x)
        return self.decode(quant), diff
