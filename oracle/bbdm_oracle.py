"""CPU oracle for the BBDM hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain-PyTorch (fp32, CPU or any device) *restatement* of the reference's
Brownian-Bridge scheduler and denoising UNet, written functionally over a reference-layout
``state_dict``.  It exists so that

  * ``tests/`` can check the HIP path against the reference's algorithm on the GPU box, where
    ``/root/reference`` does not exist,
  * ``bench.py`` can time a CPU baseline (``cpu_baseline.kind == "port"``) on the box's host cores,
  * ``__graft_entry__.smoke()`` can check one tiny invocation.

Nothing under ``bbdm_amd/`` may import it.  It never dispatches to the HIP library.

Parity pin: the reference ships no tests or golden vectors, so the pin is the
reference module itself: ``oracle/make_golden.py`` imports ``/root/reference`` in the build
container, runs the real ``BrownianBridgeModel`` / ``UNetModel`` and commits the resulting
input/output vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file
against those vectors (and directly against the live reference when it is mounted).

Each function cites the reference lines (relative to /root/reference) it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]

_UNET = "model/BrownianBridge/base/modules/diffusionmodules/openaimodel.py"
_UTIL = "model/BrownianBridge/base/modules/diffusionmodules/util.py"
_BB = "model/BrownianBridge/BrownianBridgeModel.py"


# --------------------------------------------------------------------------------------
# Brownian-Bridge schedule and scheduler arithmetic
# --------------------------------------------------------------------------------------
def make_schedule(num_timesteps: int, mt_type: str = "linear", max_var: float = 1.0,
                  skip_sample: bool = True, sample_type: str = "linear", sample_step: int = 200):
    """Schedule buffers + sampling step table.  Follows BrownianBridgeModel.py:42-79.

    Returns (buffers: dict name -> float32 tensor[T], steps: int64 tensor).
    """
    T = num_timesteps
    if mt_type == "linear":
        m = np.linspace(0.001, 0.999, T)
    elif mt_type == "sin":
        m = 1.0075 ** np.linspace(0, T, T)
        m = m / m[-1]
        m[-1] = 0.999
    else:
        raise NotImplementedError
    m_prev = np.append(0, m[:-1])
    var = 2.0 * (m - m ** 2) * max_var
    var_prev = np.append(0.0, var[:-1])
    var_t_prev = var - var_prev * ((1.0 - m) / (1.0 - m_prev)) ** 2
    post = var_t_prev * var_prev / var
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    bufs = {
        "m_t": f32(m), "m_tminus": f32(m_prev), "variance_t": f32(var),
        "variance_tminus": f32(var_prev), "variance_t_tminus": f32(var_t_prev),
        "posterior_variance_t": f32(post),
    }
    if skip_sample:
        if sample_type == "linear":
            mid = torch.arange(T - 1, 1, step=-((T - 1) / (sample_step - 2))).long()
            steps = torch.cat((mid, torch.tensor([1, 0], dtype=torch.long)), dim=0)
        elif sample_type == "cosine":
            s = np.linspace(start=0, stop=T, num=sample_step + 1)
            s = (np.cos(s / T * np.pi) + 1.0) / 2.0 * T
            steps = torch.from_numpy(s)
        else:
            steps = None
    else:
        steps = torch.arange(T - 1, -1, -1)
    return bufs, steps


def _gather(table: Tensor, t: Tensor, ndim: int) -> Tensor:
    """model/utils.py:4-7 (extract)."""
    return table.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_sample(bufs, x0: Tensor, y: Tensor, t: Tensor, noise: Tensor, objective: str = "grad"):
    """BrownianBridgeModel.py:128-146 -> (x_t, objective_target)."""
    m = _gather(bufs["m_t"], t, x0.dim())
    sig = torch.sqrt(_gather(bufs["variance_t"], t, x0.dim()))
    if objective == "grad":
        target = m * (y - x0) + sig * noise
    elif objective == "noise":
        target = noise
    elif objective == "ysubx":
        target = y - x0
    else:
        raise NotImplementedError
    return (1.0 - m) * x0 + m * y + sig * noise, target


def predict_x0(bufs, x_t: Tensor, y: Tensor, t: Tensor, pred: Tensor, objective: str = "grad"):
    """BrownianBridgeModel.py:148-160."""
    if objective == "grad":
        return x_t - pred
    if objective == "noise":
        m = _gather(bufs["m_t"], t, x_t.dim())
        sig = torch.sqrt(_gather(bufs["variance_t"], t, x_t.dim()))
        return (x_t - m * y - sig * pred) / (1.0 - m)
    if objective == "ysubx":
        return y - pred
    raise NotImplementedError


def p_sample_update(bufs, steps: Tensor, i: int, x_t: Tensor, y: Tensor, pred: Tensor,
                    noise: Optional[Tensor], objective: str = "grad", eta: float = 1.0,
                    clip_denoised: bool = False):
    """The arithmetic of BrownianBridgeModel.py:171-201 *after* the UNet call.

    ``pred`` is denoise_fn(x_t, steps[i]).  Returns (x_{t-1}, x0_recon).
    """
    b = x_t.shape[0]
    t = torch.full((b,), int(steps[i]), device=x_t.device, dtype=torch.long)
    x0r = predict_x0(bufs, x_t, y, t, pred, objective)
    if clip_denoised:
        x0r = x0r.clamp(-1.0, 1.0)
    if int(steps[i]) == 0:
        return x0r, x0r
    nt = torch.full((b,), int(steps[i + 1]), device=x_t.device, dtype=torch.long)
    m_t = _gather(bufs["m_t"], t, x_t.dim())
    m_nt = _gather(bufs["m_t"], nt, x_t.dim())
    v_t = _gather(bufs["variance_t"], t, x_t.dim())
    v_nt = _gather(bufs["variance_t"], nt, x_t.dim())
    s2 = (v_t - v_nt * (1.0 - m_t) ** 2 / (1.0 - m_nt) ** 2) * v_nt / v_t
    s = torch.sqrt(s2) * eta
    mean = (1.0 - m_nt) * x0r + m_nt * y + torch.sqrt((v_nt - s2) / v_t) * (x_t - (1.0 - m_t) * x0r - m_t * y)
    return mean + s * noise, x0r


def bb_loss(target: Tensor, pred: Tensor, loss_type: str = "l1") -> Tensor:
    """BrownianBridgeModel.py:114-119."""
    if loss_type == "l1":
        return (target - pred).abs().mean()
    if loss_type == "l2":
        return F.mse_loss(target, pred)
    raise NotImplementedError


# --------------------------------------------------------------------------------------
# UNet (functional, over a reference-layout state_dict)
# --------------------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """util.py:151-171 (cos first, then sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    ang = t[:, None].float() * freqs[None]
    e = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        e = torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1)
    return e


def _gn(sd: StateDict, key: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    """GroupNorm32(32, C): util.py:199-216."""
    w = sd[key + ".weight"]          # fp32 weights: x.float() as in the reference; fp64 weights (test probe): stay in fp64
    return F.group_norm(x.to(w.dtype), 32, w, sd[key + ".bias"], eps).type(x.dtype)


def _conv(sd: StateDict, key: str, x: Tensor, padding: int = 0, stride: int = 1) -> Tensor:
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def _lin(sd: StateDict, key: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def resblock(sd: StateDict, p: str, x: Tensor, emb: Tensor, up: bool, down: bool,
             scale_shift: bool) -> Tensor:
    """ResBlock._forward, openaimodel.py:258-278 (ctor :182-244)."""
    h = F.silu(_gn(sd, p + "in_layers.0", x))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(sd, p + "in_layers.2", h, padding=1)
    e = _lin(sd, p + "emb_layers.1", F.silu(emb))[:, :, None, None]
    if scale_shift:
        sc, sh = torch.chunk(e, 2, dim=1)
        h = _gn(sd, p + "out_layers.0", h) * (1 + sc) + sh
    else:
        h = _gn(sd, p + "out_layers.0", h + e)
    h = _conv(sd, p + "out_layers.3", F.silu(h), padding=1)   # Dropout(p=0) is the identity
    if (p + "skip_connection.weight") in sd:
        w = sd[p + "skip_connection.weight"]
        x = F.conv2d(x, w, sd[p + "skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def attention_block(sd: StateDict, p: str, x: Tensor, n_heads: int, new_order: bool = False) -> Tensor:
    """AttentionBlock._forward :321-327 + QKVAttentionLegacy :359-375 / QKVAttention :398-413."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + "norm", xf), sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    T = xf.shape[-1]
    ch = c // n_heads
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(b * n_heads, ch, T) for z in (q, k, v))
    else:
        q, k, v = qkv.reshape(b * n_heads, 3 * ch, T).split(ch, dim=1)
    s = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, T)
    a = F.conv1d(a, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return (xf + a).reshape(b, c, hh, ww)


def _cross_attention(sd: StateDict, p: str, x: Tensor, context: Optional[Tensor], heads: int) -> Tensor:
    """CrossAttention.forward, model/BrownianBridge/base/modules/attention.py:170-194.  x: [b, n, c] tokens;
    context: [b, c', h, w] image or None (= self-attention)."""
    q = F.linear(x, sd[p + "to_q.weight"])
    ctx = x if context is None else context.flatten(2).transpose(1, 2)            # 'b c h w -> b (h w) c'
    k = F.linear(ctx, sd[p + "to_k.weight"])
    v = F.linear(ctx, sd[p + "to_v.weight"])
    b, n, inner = q.shape
    d = inner // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    out = torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def spatial_transformer(sd: StateDict, p: str, x: Tensor, context: Optional[Tensor], heads: int, depth: int) -> Tensor:
    """SpatialTransformer.forward + BasicTransformerBlock._forward + FeedForward/GEGLU (attention.py:38-64,196-263)."""
    b, c, hh, ww = x.shape
    h = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    h = F.conv2d(h, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    t = h.flatten(2).transpose(1, 2)                                               # 'b c h w -> b (h w) c'
    for i in range(depth):
        q = f"{p}transformer_blocks.{i}."
        ln = lambda name, z: F.layer_norm(z, z.shape[-1:], sd[q + name + ".weight"], sd[q + name + ".bias"], 1e-5)
        t = _cross_attention(sd, q + "attn1.", ln("norm1", t), None, heads) + t
        t = _cross_attention(sd, q + "attn2.", ln("norm2", t), context, heads) + t
        g = F.linear(ln("norm3", t), sd[q + "ff.net.0.proj.weight"], sd[q + "ff.net.0.proj.bias"])
        a, gate = g.chunk(2, dim=-1)
        t = F.linear(a * F.gelu(gate), sd[q + "ff.net.2.weight"], sd[q + "ff.net.2.bias"]) + t
    h = t.transpose(1, 2).reshape(b, -1, hh, ww)
    return F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"]) + x


class UNetSpec:
    """The UNetParams keys that shape the graph (openaimodel.py:446-473)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True, condition_key="concat"):
        assert dims == 2 and num_classes is None and n_embed is None
        self.use_spatial_transformer = use_spatial_transformer
        self.transformer_depth = transformer_depth
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = tuple(channel_mult)
        self.conv_resample = conv_resample
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample
        self.use_scale_shift_norm = use_scale_shift_norm
        self.resblock_updown = resblock_updown
        self.use_new_attention_order = use_new_attention_order
        self.condition_key = condition_key

    def layout(self):
        """Walk the constructor loops of openaimodel.py:518-685 and list, per block, the sub-layers.

        Returns (input_blocks, middle, output_blocks); each block is a list of tuples
        ('conv',) | ('res', up, down) | ('attn', heads) | ('down_conv',) | ('down_pool',) | ('up', has_conv).
        """
        mc = self.model_channels
        heads_of = lambda ch, default: (default if self.num_head_channels == -1 else ch // self.num_head_channels)
        # with use_spatial_transformer every AttentionBlock site holds a SpatialTransformer (openaimodel.py:556-565,610-618,
        # 658-666) whose head count is ch // num_head_channels when that is set, else num_heads
        kind_attn = "st" if self.use_spatial_transformer else "attn"
        inp: List[List[tuple]] = [[("conv",)]]
        ch, ds = mc, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks):
                blk = [("res", False, False)]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    blk.append((kind_attn, heads_of(ch, self.num_heads)))
                inp.append(blk)
            if level != len(self.channel_mult) - 1:
                if self.resblock_updown:
                    inp.append([("res", False, True)])
                else:
                    inp.append([("down_conv",) if self.conv_resample else ("down_pool",)])
                ds *= 2
        mid = [("res", False, False), (kind_attn, heads_of(ch, self.num_heads)), ("res", False, False)]
        out: List[List[tuple]] = []
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(self.num_res_blocks + 1):
                blk = [("res", False, False)]
                ch = mc * mult
                if ds in self.attention_resolutions:
                    # openaimodel.py:662 passes num_heads_upsample, but AttentionBlock prefers
                    # num_head_channels when it is not -1 (openaimodel.py:298-304)
                    blk.append((kind_attn, heads_of(ch, self.num_heads if self.use_spatial_transformer
                                                    else self.num_heads_upsample)))
                if level and i == self.num_res_blocks:
                    blk.append(("res", True, False) if self.resblock_updown else ("up", self.conv_resample))
                    ds //= 2
                out.append(blk)
        return inp, mid, out


def _run_block(sd, spec: UNetSpec, prefix: str, blk, h, emb, context=None):
    for j, layer in enumerate(blk):
        p = f"{prefix}{j}."
        kind = layer[0]
        if kind == "conv":
            h = _conv(sd, p[:-1], h, padding=1)
        elif kind == "res":
            h = resblock(sd, p, h, emb, up=layer[1], down=layer[2], scale_shift=spec.use_scale_shift_norm)
        elif kind == "attn":
            h = attention_block(sd, p, h, layer[1], spec.use_new_attention_order)
        elif kind == "st":
            h = spatial_transformer(sd, p, h, context, layer[1], spec.transformer_depth)
        elif kind == "down_conv":
            h = _conv(sd, p + "op", h, padding=1, stride=2)
        elif kind == "down_pool":
            h = F.avg_pool2d(h, 2, 2)
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if layer[1]:
                h = _conv(sd, p + "conv", h, padding=1)
        else:
            raise ValueError(kind)
    return h


def unet_forward(sd: StateDict, spec: UNetSpec, x: Tensor, timesteps: Tensor,
                 context: Optional[Tensor] = None, prefix: str = "") -> Tensor:
    """UNetModel.forward, openaimodel.py:721-759.  ``sd`` keys are ``prefix + <UNetModel key>``."""
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    inp, mid, out = spec.layout()
    emb = timestep_embedding(timesteps, spec.model_channels)
    emb = emb.to(sd["time_embed.0.weight"].dtype)     # no-op in fp32; lets the tests run the oracle in fp64 as a conditioning probe
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", emb)))
    if spec.condition_key != "nocond":
        x = torch.cat([x, context], dim=1)
    h = x
    hs = []
    for i, blk in enumerate(inp):
        h = _run_block(sd, spec, f"input_blocks.{i}.", blk, h, emb, context)
        hs.append(h)
    h = _run_block(sd, spec, "middle_block.", mid, h, emb, context)
    for i, blk in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, spec, f"output_blocks.{i}.", blk, h, emb, context)
    h = F.silu(_gn(sd, "out.0", h))
    return _conv(sd, "out.2", h, padding=1)


# --------------------------------------------------------------------------------------
# Whole-model helpers (mirror BrownianBridgeModel.forward / p_sample / p_sample_loop)
# --------------------------------------------------------------------------------------
class OracleBBDM:
    """Stateless-ish driver: schedule + UNet state_dict, CPU fp32.  Mirrors BrownianBridgeModel.py:15-225."""

    def __init__(self, sd: StateDict, unet_spec: UNetSpec, num_timesteps=1000, mt_type="linear", max_var=1.0,
                 eta=1.0, skip_sample=True, sample_type="linear", sample_step=200, loss_type="l1",
                 objective="grad", unet_prefix: str = "denoise_fn."):
        self.sd = sd
        self.spec = unet_spec
        self.prefix = unet_prefix
        self.bufs, self.steps = make_schedule(num_timesteps, mt_type, max_var, skip_sample, sample_type, sample_step)
        self.eta, self.loss_type, self.objective = eta, loss_type, objective
        self._sd_unet = {k[len(unet_prefix):]: v for k, v in sd.items() if k.startswith(unet_prefix)}

    def denoise(self, x_t, t, context):
        return unet_forward(self._sd_unet, self.spec, x_t, t, context)

    def _ctx(self, y, context):
        if self.spec.condition_key == "nocond":
            return None
        return y if context is None else context

    @torch.no_grad()
    def p_sample(self, x_t, y, context, i, clip_denoised=False, noise=None):
        t = torch.full((x_t.shape[0],), int(self.steps[i]), dtype=torch.long, device=x_t.device)
        pred = self.denoise(x_t, t, context)
        if noise is None and int(self.steps[i]) != 0:
            noise = torch.randn_like(x_t)
        return p_sample_update(self.bufs, self.steps, i, x_t, y, pred, noise, self.objective, self.eta,
                               clip_denoised)

    @torch.no_grad()
    def p_sample_loop(self, y, context=None, clip_denoised=True, noises: Optional[Sequence[Tensor]] = None):
        context = self._ctx(y, context)
        img = y
        for i in range(len(self.steps)):
            img, _ = self.p_sample(img, y, context, i, clip_denoised, None if noises is None else noises[i])
        return img

    def p_losses(self, x0, y, context, t, noise):
        x_t, target = q_sample(self.bufs, x0, y, t, noise, self.objective)
        pred = self.denoise(x_t, t, self._ctx(y, context))
        loss = bb_loss(target, pred, self.loss_type)
        return loss, {"loss": loss, "x0_recon": predict_x0(self.bufs, x_t, y, t, pred, self.objective)}
