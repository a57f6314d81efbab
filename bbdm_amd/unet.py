"""UNetModel -- the BBDM denoiser executed by hand-written HIP kernels (libbbdm_hip.so).

Mirror of the reference's ``UNetModel`` (model/BrownianBridge/base/modules/diffusionmodules/openaimodel.py:416-759):
same constructor keywords (``UNetModel(**vars(UNetParams))``, BrownianBridgeModel.py:40), same sub-module tree and
therefore the same ``state_dict()`` keys / shapes / dtypes and the same parameter-initialisation order under a
given seed, same ``forward(x, timesteps, context)`` contract (NCHW fp32 in, NCHW fp32 out).

The ``nn.Conv2d`` / ``nn.Linear`` / ``GroupNorm32`` children are *parameter holders* (so the runner's
``weights_init`` -- runners/utils.py:35-45, which dispatches on class names -- EMA, the optimizer, DDP and
``load_state_dict`` behave exactly as with the reference); their own ``forward`` is never used.  ``forward`` here
compiles, per input shape, a static list of C-ABI calls over pre-allocated NHWC fp32 buffers and replays it.

There is no PyTorch fallback: without the HIP library, or on a non-GPU tensor, ``forward`` raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

__all__ = ["UNetModel"]


class _NoVersion:
    """Stands in for the version of a tensor that has no version counter: never equal to anything.  (A FRESH instance per read:
    container comparisons short-cut on identity.)"""

    def __eq__(self, other):
        return False

    __hash__ = object.__hash__


def _ver(t: torch.Tensor):
    """``t._version``, or a :class:`_NoVersion` for inference tensors (``torch.inference_mode()``: no version counter).  The version
    only tracks writes made through torch ops on ``t`` or its views -- a raw-pointer kernel or a DLPack consumer does not bump it --
    so every cache keyed on it must tolerate a miss: a key containing a ``_NoVersion`` never matches, i.e. the copy / re-pack runs."""
    if t.is_inference():
        return _NoVersion()
    try:
        return t._version
    except RuntimeError:
        return _NoVersion()


# --------------------------------------------------------------------------------------------------------------
# parameter-holder modules (names follow the reference so that class-name based init hooks match)
# --------------------------------------------------------------------------------------------------------------
class GroupNorm32(nn.GroupNorm):
    """util.py:214-216."""


def _zero_(module: nn.Module) -> nn.Module:
    """util.py:174-180 (zero_module)."""
    for p in module.parameters():
        p.detach().zero_()
    return module


def _no_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a parameter holder of bbdm_amd.UNetModel; call the UNetModel")


class TimestepEmbedSequential(nn.Sequential):
    """openaimodel.py:75-90 (container only)."""
    forward = _no_forward


class Upsample(nn.Module):
    """openaimodel.py:93-121."""

    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=1)

    forward = _no_forward


class Downsample(nn.Module):
    """openaimodel.py:137-163."""

    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    forward = _no_forward


class ResBlock(nn.Module):
    """openaimodel.py:166-278."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False,
                 up=False, down=False):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        self.up, self.down = up, down
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False), Upsample(channels, False)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False), Downsample(channels, False)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(
            nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        _zero_(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)

    forward = _no_forward


class AttentionBlock(nn.Module):
    """openaimodel.py:281-327."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            self.num_heads = channels // num_head_channels
        self.use_new_attention_order = use_new_attention_order
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = _zero_(nn.Conv1d(channels, channels, 1))

    forward = _no_forward


class CrossAttention(nn.Module):
    """model/BrownianBridge/base/modules/attention.py:153-194 (parameter holder)."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale, self.heads, self.dim_head = dim_head ** -0.5, heads, dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))

    forward = _no_forward


class GEGLU(nn.Module):
    """attention.py:38-46."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    forward = _no_forward


class FeedForward(nn.Module):
    """attention.py:48-64."""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        project_in = nn.Sequential(nn.Linear(dim, inner_dim), nn.GELU()) if not glu else GEGLU(dim, inner_dim)
        self.net = nn.Sequential(project_in, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))

    forward = _no_forward


class BasicTransformerBlock(nn.Module):
    """attention.py:196-219 (construction order = the reference's: attn1, ff, attn2, norm1..3)."""

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    forward = _no_forward


class SpatialTransformer(nn.Module):
    """attention.py:222-263: GroupNorm(eps 1e-6) -> 1x1 conv -> depth x BasicTransformerBlock over the pixel tokens (self-
    attention, cross-attention to the context image's pixels, GEGLU feed-forward) -> zero-initialised 1x1 conv + input."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None):
        super().__init__()
        self.in_channels, self.n_heads, self.d_head = in_channels, n_heads, d_head
        inner_dim = n_heads * d_head
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim)
             for _ in range(depth)])
        self.proj_out = _zero_(nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))

    forward = _no_forward


# --------------------------------------------------------------------------------------------------------------
# small helpers for the executor
# --------------------------------------------------------------------------------------------------------------
class _Buf:
    """A device allocation whose size is only known once the whole plan has been emitted."""
    __slots__ = ("numel", "tensor")

    def __init__(self, numel=0):
        self.numel, self.tensor = numel, None


class _View:
    """An NHWC fp32 activation: (owning buffer, element offset, pitch, N, H, W, C).  Resolves to a device pointer."""
    __slots__ = ("buf", "off", "ld", "N", "H", "W", "C")

    def __init__(self, buf, off, ld, N, H, W, C):
        self.buf, self.off, self.ld, self.N, self.H, self.W, self.C = buf, off, ld, N, H, W, C

    def resolve(self):
        return self.buf.tensor.data_ptr() + 4 * self.off


class _TensorRef:
    """Pointer into a tensor owned by the plan (+ byte offset)."""
    __slots__ = ("t", "byte_off")

    def __init__(self, t, byte_off=0):
        self.t, self.byte_off = t, byte_off

    def resolve(self):
        t = self.t.t if isinstance(self.t, _LateTensor) else self.t
        return t.data_ptr() + self.byte_off


class _ParamRef:
    """Pointer to a model parameter, re-resolved whenever any parameter storage moves (EMA swaps ``param.data``,
    runners/base/EMA.py:31-43; ``load_state_dict`` copies in place and keeps pointers)."""
    __slots__ = ("p",)

    def __init__(self, p):
        self.p = p

    def resolve(self):
        p = self.p
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("bbdm_amd: parameters must be contiguous fp32")
        return p.data_ptr()


class _LateTensor:
    """A workspace whose size is only known after emission; ``t`` is assigned before the first bind."""
    __slots__ = ("t",)

    def __init__(self):
        self.t = None

    def resolve(self):
        return self.t.data_ptr()


class _LateInt:
    __slots__ = ("v",)

    def __init__(self):
        self.v = 0

    def resolve(self):
        return self.v


class _OpName(str):
    """Plan-level op name (what bench.py / the tests key on) that binds to a different C entry point: the Winograd tile
    GEMMs stay "bbdm_winograd_gemm_f32" in the op list whether they run on the f32 or on the bf16x3 kernel."""
    entry: str = ""

    def __new__(cls, name, entry):
        o = super().__new__(cls, name)
        o.entry = entry
        return o


class _GnPre(tuple):
    """Fused-producer arguments of a Winograd input transform that forms the GroupNorm coefficients itself
    (bbdm_winograd_input_bf3p_gn_f32): (stats reference, None, C, silu) in the positions of (pre_scale, pre_bias, pre_ld, pre_silu),
    the GroupNorm's parameters in ``tail`` (appended after CinPad)."""
    tail: tuple = ()
    h2 = None          # reference of the device float that bounds the producer's output (see _Pre), or None

    def __new__(cls, head, tail, h2=None):
        o = super().__new__(cls, head)
        o.tail = tuple(tail)
        o.h2 = h2
        return o


class _Pre(tuple):
    """(pre_scale, pre_bias, pre_ld, pre_silu) of a fused producer -- all None / 0 when the tensor was materialised -- plus ``h2``: the
    reference of a device float >= max |value| of what the consumer convolves (a GroupNorm output is bounded by its coefficients alone:
    csrc/groupnorm.hip, h2_gn_bounds_kernel), which lets the consumer's tile GEMMs run on the fp16-pair planes (csrc/h2_split.h)."""
    h2 = None

    def __new__(cls, head, h2=None):
        o = super().__new__(cls, head)
        o.h2 = h2
        return o


def _round4(c):
    return (c + 3) // 4 * 4


def _zero_on(t: torch.Tensor, stream):
    """``t.zero_()`` enqueued on the raw HIP stream ``stream`` -- the stream the packing launches that follow are given.  (The training
    plans re-pack their data-gradient operands on the plan's SECOND stream: a plain ``zero_()`` would go to torch's current stream and
    race with the maximum those launches accumulate into ``t``.)"""
    if t.is_cuda and stream:
        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=t.device)):
            t.zero_()
    else:
        t.zero_()


class _PackedConv:
    """Packed copy of one conv weight, refreshed when the parameter storage or version changes
    (EMA swaps ``param.data`` without bumping ``_version`` -- runners/base/EMA.py:31-43 -- so both are keyed)."""

    def __init__(self, weight: nn.Parameter, bias: Optional[nn.Parameter], cin_pad: int):
        self.weight, self.bias = weight, bias
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.ks = weight.shape[2] if weight.dim() == 4 else 1      # Conv1d k=1: [O, I, 1] == [O, I, 1, 1] in memory
        self.cin_pad = cin_pad
        n = _lib.load().bbdm_conv_packed_floats(self.cout, cin_pad, self.ks)
        self.packed = torch.empty(n, dtype=torch.float32, device=weight.device)
        self.packed.cin_true = self.cin          # algorithmic (unpadded) input channels, for flop accounting
        self.key = None

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise RuntimeError("bbdm_amd: conv weights must be contiguous fp32")
            _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), self.packed.data_ptr(), self.cout, self.cin,
                      self.cin_pad, self.ks, stream)
            self.key = key


class _PackedConvBf3(_PackedConv):
    """A 1x1 conv / Linear weight split into the three bf16 planes of csrc/gemm_bf3.hip (``packed``); the fp32 packing is
    kept as the intermediate."""

    def __init__(self, weight, bias, cin_pad):
        super().__init__(weight, bias, cin_pad)
        self.packed_f32 = self.packed
        nh = _lib.load().bbdm_gemm_bf3_packed_halfs(1, cin_pad, self.cout)
        self.packed = torch.empty(nh, dtype=torch.int16, device=weight.device)
        self.packed.cin_true = self.cin

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise RuntimeError("bbdm_amd: conv weights must be contiguous fp32")
            _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), self.packed_f32.data_ptr(), self.cout, self.cin,
                      self.cin_pad, 1, stream)
            _lib.call("bbdm_gemm_bf3_pack_f32", self.packed_f32.data_ptr(), self.packed.data_ptr(), 1, self.cin_pad, self.cout,
                      stream)
            self.key = key


class _PackedConvBf3q(_PackedConv):
    """A 1x1 conv / Linear weight as the B planes of csrc/gemm_bf3p.hip (``packed``; fragment-unit layout): the operand of
    bbdm_conv1x1_bf3q_f32, the pipelined kernel that reads the fp32 activation as it lies in HBM."""

    def __init__(self, weight, bias, cin_pad):
        super().__init__(weight, bias, cin_pad)
        self.packed_f32 = self.packed
        self.packed = torch.empty(_lib.load().bbdm_gemm_bf3p_b_bytes(1, cin_pad, self.cout), dtype=torch.uint8, device=weight.device)
        self.packed.cin_true = self.cin

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise RuntimeError("bbdm_amd: conv weights must be contiguous fp32")
            _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), self.packed_f32.data_ptr(), self.cout, self.cin,
                      self.cin_pad, 1, stream)
            _lib.call("bbdm_gemm_bf3p_pack_b_f32", self.packed_f32.data_ptr(), self.packed.data_ptr(), 1, self.cin_pad, self.cout,
                      stream)
            self.key = key


class _PackedConvH2q(_PackedConv):
    """A 1x1 conv / Linear weight as the fp16-pair B planes of csrc/gemm_bf3p.hip under the scale of its exact maximum (``ubound``): the
    operand of bbdm_conv1x1_h2q_f32."""

    def __init__(self, weight, bias, cin_pad):
        super().__init__(weight, bias, cin_pad)
        self.packed_f32 = self.packed
        self.packed = torch.empty(_lib.load().bbdm_gemm_h2p_b_bytes(1, cin_pad, self.cout), dtype=torch.uint8, device=weight.device)
        self.packed.cin_true = self.cin
        self.ubound = torch.zeros(1, dtype=torch.float32, device=weight.device)

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise RuntimeError("bbdm_amd: conv weights must be contiguous fp32")
            _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), self.packed_f32.data_ptr(), self.cout, self.cin,
                      self.cin_pad, 1, stream)
            _zero_on(self.ubound, stream)
            _lib.call("bbdm_absmax_f32", self.packed_f32.data_ptr(), self.packed_f32.numel(), self.ubound.data_ptr(), stream)
            _lib.call("bbdm_gemm_h2p_pack_b_f32", self.packed_f32.data_ptr(), self.packed.data_ptr(), self.ubound.data_ptr(), 1.0, 1,
                      self.cin_pad, self.cout, stream)
            self.key = key


class _RowL1Gain:
    """(max over rows of sum |W[row, :]|, max |bias|) of a 1x1 conv / Linear as two device floats, refreshed with the weights: what
    turns a bound of the layer's input into a bound of its output (csrc/groupnorm.hip: h2_rowl1_kernel / bbdm_h2_affine_bound_f32)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.gain = torch.zeros(2, dtype=torch.float32, device=weight.device)
        self.key = None

    def refresh(self, stream):
        w, b = self.weight, self.bias
        key = (w.data_ptr(), _ver(w), None if b is None else (b.data_ptr(), _ver(b)))
        if key != self.key:
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise RuntimeError("bbdm_amd: conv weights must be contiguous fp32")
            _lib.call("bbdm_h2_rowl1_f32", w.data_ptr(), None if b is None else b.data_ptr(), w.shape[0], w[0].numel(),
                      self.gain.data_ptr(), stream)
            self.key = key


class _PackedDgradBf3:
    """The transposed weight of a 1x1 conv / Linear in the three-bf16-plane layout of csrc/gemm_bf3.hip: the data gradient
    dX = dY W as one more fp32-accurate GEMM on the BF16 matrix core (``packed``; the fp32 dgrad packing is the intermediate)."""

    def __init__(self, weight: nn.Parameter, cout_in: int, planes=False):
        """``planes``: the B planes of csrc/gemm_bf3p.hip (for bbdm_conv1x1_bf3q_f32) instead of gemm_bf3.hip's layout; "h": the
        fp16-pair planes under the weights' exact maximum ``ubound`` (bbdm_conv1x1_h2q_f32; round 6)."""
        self.weight = weight
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.cout_in, self.planes = cout_in, planes
        lib = _lib.load()
        self.packed_f32 = torch.empty(lib.bbdm_conv_packed_dgrad_floats(self.cout, self.cin, cout_in, 1), dtype=torch.float32,
                                      device=weight.device)
        self.ubound = torch.zeros(1, dtype=torch.float32, device=weight.device) if planes == "h" else None
        if planes == "h":
            self.packed = torch.empty(lib.bbdm_gemm_h2p_b_bytes(1, cout_in, self.cin), dtype=torch.uint8, device=weight.device)
        elif planes:
            self.packed = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(1, cout_in, self.cin), dtype=torch.uint8, device=weight.device)
        else:
            self.packed = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(1, cout_in, self.cin), dtype=torch.int16, device=weight.device)
        self.key = None

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            _lib.call("bbdm_conv_pack_weight_dgrad_f32", w.data_ptr(), self.packed_f32.data_ptr(), self.cout, self.cin,
                      self.cout_in, 1, stream)
            if self.planes == "h":
                _zero_on(self.ubound, stream)
                _lib.call("bbdm_absmax_f32", self.packed_f32.data_ptr(), self.packed_f32.numel(), self.ubound.data_ptr(), stream)
                _lib.call("bbdm_gemm_h2p_pack_b_f32", self.packed_f32.data_ptr(), self.packed.data_ptr(), self.ubound.data_ptr(), 1.0, 1,
                          self.cout_in, self.cin, stream)
            else:
                _lib.call("bbdm_gemm_bf3p_pack_b_f32" if self.planes else "bbdm_gemm_bf3_pack_f32", self.packed_f32.data_ptr(),
                          self.packed.data_ptr(), 1, self.cout_in, self.cin, stream)
            self.key = key


class _PackedDgrad:
    """Packed transposed + flipped copy of a conv weight: the forward kernel run with it computes the data gradient."""

    def __init__(self, weight: nn.Parameter, cout_in: int):
        self.weight = weight
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.ks = weight.shape[2] if weight.dim() == 4 else 1
        self.cout_in = cout_in
        n = _lib.load().bbdm_conv_packed_dgrad_floats(self.cout, self.cin, cout_in, self.ks)
        self.packed = torch.empty(n, dtype=torch.float32, device=weight.device)
        self.key = None

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            _lib.call("bbdm_conv_pack_weight_dgrad_f32", w.data_ptr(), self.packed.data_ptr(), self.cout, self.cin,
                      self.cout_in, self.ks, stream)
            self.key = key


class _PackedWinograd:
    """G g G^T of one 3x3 conv weight in the batched-GEMM layout (``dgrad``: of the data-gradient convolution).  With
    ``bf3`` the fp32 buffer is additionally split into the three bf16 planes csrc/gemm_bf3.hip takes; ``packed`` is then
    that buffer (the op binds to bbdm_winograd_gemm_bf3_f32).  ``bf3 == "p"``: the planes in the fragment-unit layout of
    csrc/gemm_bf3p.hip, whose A operand the input transform writes pre-split (bbdm_winograd_input_bf3p_f32 / _gemm_bf3p_f32)."""

    def __init__(self, weight: nn.Parameter, bias: Optional[nn.Parameter], in_pad: int, m: int, dgrad: bool = False,
                 bf3: bool = False, phases: bool = False):
        self.weight, self.bias, self.dgrad, self.m, self.bf3 = weight, bias, dgrad, m, bf3
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.ks, self.in_pad = 3, in_pad
        lib = _lib.load()
        # ``phases``: the layer is conv3x3(nearest x2 (x)); packed are its four phase filters, a conv Cin -> 4 Cout on x itself
        # (bbdm_upsample_phase_weights_f32, BBDM_CONV_OUT_PHASES)
        assert not (phases and dgrad)
        self.phases = phases
        self.w4 = torch.empty(4 * self.cout, self.cin, 3, 3, dtype=torch.float32, device=weight.device) if phases else None
        self.out_ch = self.cin if dgrad else (4 * self.cout if phases else self.cout)
        n = lib.bbdm_winograd_packed_floats(m, self.out_ch, in_pad)
        # (the planes of gemm_bf3p.hip are written directly from the weights: no fp32 G g G^T tensor -- 4x the weights at m = 4 -- is kept)
        self.fused_planes = bf3 == "p" and not phases and in_pad % 16 == 0
        # bf3 == "h": two fp16 planes under the scale of max |U| (csrc/h2_split.h); the fp32 G g G^T tensor only exists while packing
        self.packed_f32 = None if (self.fused_planes or bf3 == "h") else torch.empty(n, dtype=torch.float32, device=weight.device)
        self._n_f32 = n
        self.ubound = torch.zeros(1, dtype=torch.float32, device=weight.device) if bf3 == "h" else None
        if bf3 == "h":
            self.packed = torch.empty(lib.bbdm_gemm_h2p_b_bytes(wino_planes(m), in_pad, self.out_ch), dtype=torch.uint8,
                                      device=weight.device)
        elif bf3 == "p":
            self.packed = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(wino_planes(m), in_pad, self.out_ch), dtype=torch.uint8,
                                      device=weight.device)
        elif bf3:
            nh = lib.bbdm_gemm_bf3_packed_halfs(wino_planes(m), in_pad, self.out_ch)
            self.packed = torch.empty(nh, dtype=torch.int16, device=weight.device)
        else:
            self.packed = self.packed_f32
        self.packed.cin_true = self.cout if dgrad else self.cin
        self.key = None

    def refresh(self, stream):
        w = self.weight
        key = (w.data_ptr(), _ver(w))
        if key != self.key:
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise RuntimeError("bbdm_amd: conv weights must be contiguous fp32")
            if self.bf3 == "h":
                # fp16-pair planes of U 2^e: e from ``ubound`` = the filter's largest tap (one small pass over the weights) times the gain
                # of G . G^T, the factor the tile GEMM applies to the same pointer -- G g G^T goes straight into the planes
                _zero_on(self.ubound, stream)
                if self.phases:
                    _lib.call("bbdm_upsample_phase_weights_f32", w.data_ptr(), self.w4.data_ptr(), self.cout, self.cin, stream)
                    _lib.call("bbdm_absmax_f32", self.w4.data_ptr(), self.w4.numel(), self.ubound.data_ptr(), stream)
                    tmp = torch.empty(self._n_f32, dtype=torch.float32, device=w.device)
                    _lib.call("bbdm_winograd_pack_weight_f32", self.m, self.w4.data_ptr(), tmp.data_ptr(), 4 * self.cout, self.cin,
                              self.in_pad, 0, stream)
                    _lib.call("bbdm_gemm_h2p_pack_b_f32", tmp.data_ptr(), self.packed.data_ptr(), self.ubound.data_ptr(),
                              float(_lib.load().bbdm_winograd_g_gain(self.m)), wino_planes(self.m), self.in_pad, self.out_ch, stream)
                else:
                    _lib.call("bbdm_absmax_f32", w.data_ptr(), w.numel(), self.ubound.data_ptr(), stream)
                    _lib.call("bbdm_winograd_pack_weight_h2p_f32", self.m, w.data_ptr(), self.packed.data_ptr(), self.cout, self.cin,
                              self.in_pad, 1 if self.dgrad else 0, self.ubound.data_ptr(), stream)
                self.key = key
                return
            if self.phases:
                _lib.call("bbdm_upsample_phase_weights_f32", w.data_ptr(), self.w4.data_ptr(), self.cout, self.cin, stream)
                _lib.call("bbdm_winograd_pack_weight_f32", self.m, self.w4.data_ptr(), self.packed_f32.data_ptr(), 4 * self.cout,
                          self.cin, self.in_pad, 0, stream)
            elif self.fused_planes:
                # G g G^T straight into the bf16 planes (the two launches below in one, without the fp32 tensor in between)
                _lib.call("bbdm_winograd_pack_weight_bf3p_f32", self.m, w.data_ptr(), self.packed.data_ptr(), self.cout, self.cin,
                          self.in_pad, 1 if self.dgrad else 0, stream)
                self.key = key
                return
            else:
                _lib.call("bbdm_winograd_pack_weight_f32", self.m, w.data_ptr(), self.packed_f32.data_ptr(), self.cout, self.cin,
                          self.in_pad, 1 if self.dgrad else 0, stream)
            if self.bf3:
                _lib.call("bbdm_gemm_bf3p_pack_b_f32" if self.bf3 == "p" else "bbdm_gemm_bf3_pack_f32",
                          self.packed_f32.data_ptr(), self.packed.data_ptr(), wino_planes(self.m), self.in_pad, self.out_ch, stream)
            self.key = key


def wino_planes(m: int) -> int:
    """Transform points of Winograd tile ``m``: (m + 2)^2 for F(m x m, 3x3); m = 7 is F(7x7, 2x2) on the 8-point transform."""
    return 64 if m == 7 else (m + 2) ** 2


def wino_tiles(m: int, N: int, H: int, W: int) -> int:
    """Real (unpadded) tiles of an [N, H, W] image (csrc/winograd_math.h: wino_tdim)."""
    if m == 7:
        return N * ((H + 7) // 7) * ((W + 7) // 7)
    return N * -(-H // m) * -(-W // m)


def phase_filter_tile(N: int, H: int, W: int, cin: int, cout4: int, max_m: int, small: bool, f72: bool = True) -> int:
    """Winograd tile for conv3x3(nearest x2 (x)) run as four phase filters on x [N, H, W] (Cin -> cout4 = 4 Cout).  Each phase filter
    reads 2 x 2 pixels of x, so where the layer earns the 8-point transform it runs as F(7x7, 2x2) -- 49 outputs per 64 multiplies
    instead of 36 (round 5: -17 % tile GEMM work at 64^2, -25 % at 128^2 incl. the edge tiles) -- as long as the coarser tile grid does
    not eat the gain and the output transform's 128-channel blocks lie inside one phase."""
    wl = winograd_tile(N, H, W, cin, cout4, max_m, small=small)
    if f72 and wl == 6 and (cout4 // 4) % 128 == 0 and cin % 16 == 0 and wino_tiles(7, N, H, W) <= 0.9 * wino_tiles(6, N, H, W):
        return 7
    return wl


def _tile8_ok(N: int, H: int, W: int, cin: int, cout: int, min_tiles: int = 512) -> bool:
    """Does this layer earn F(8x8, 3x3)?  Large layers on the pre-split planes: whole 16-channel chunks in, whole 128-channel blocks out,
    >= 512 tiles (fewer: 100 transform points x the weight planes become the bound), <= 10 % edge waste."""
    t8h, t8w = -(-H // 8), -(-W // 8)
    return (cin >= 128 and cin % 16 == 0 and cout % 128 == 0 and cin * cout / (cin + cout) >= 64 and N * t8h * t8w >= min_tiles
            and (8 * t8h) * (8 * t8w) <= 1.10 * H * W)


def winograd_tile(N: int, H: int, W: int, cin: int, cout: int, max_m: int = 6, small: bool = True, allow8: bool = False) -> int:
    """Output tile m of the Winograd F(m x m, 3x3) path for this layer, or 0 = direct implicit GEMM.

    Measured on MI355X (tools/wino_bench.py -> profiles/r02_wino_bench.txt): the 2.25x (m = 2) / 4x
    (m = 4) / 5.06x (m = 6) cut in MFMA work must outweigh the HBM passes of the two transforms -- (m+2)^2/m^2 x the input
    plus the same for the output -- which needs wide layers (harmonic width cin*cout/(cin+cout)) and enough tiles to fill
    the chip with 256-row GEMM tiles.  m = 6 beats m = 4 by 1.1-1.25x once there are >= ~900 8x8 tiles and the masked edge
    tiles waste <= 10 % (H, W need not be multiples of 6); on 32x32 / 16x16 latents (27 % edge waste) m = 4 stays."""
    if cin % 4 or cout % 4 or cout < 128:
        return 0
    hw = cin * cout / (cin + cout)
    # m = 8 (round 5; ``allow8``: the bf16x3 pipeline is on -- inference forward, and since the end of round 5 the training forward /
    # data gradient / Winograd-domain weight gradient, UNetModel.winograd_train8): 100 / 64 = 1.56 multiplies per output instead
    # of m = 6's 1.78 (m = 4: 2.25), and 64 / 128 / 256-pixel images tile without the 2 - 6 % edge waste of the 6-pixel grid: -14 ... -17 %
    # tile GEMM work and transformed bytes (-31 % against m = 4 on 32^2 maps).  Its fp32 error is ~7x m = 6's
    # (tests/test_winograd_math_cpu.py): UNetModel.winograd = 8 opts in.
    # (``allow8``: False / True = the default threshold of 512 tiles, or the threshold itself -- UNetModel.winograd8_min_tiles)
    if allow8 and max_m >= 8 and _tile8_ok(N, H, W, cin, cout, 512 if allow8 is True else int(allow8)):
        return 8
    t6h, t6w = -(-H // 6), -(-W // 6)
    if (max_m >= 6 and cin >= 128 and hw >= 64 and N * t6h * t6w >= 900
            and (6 * t6h) * (6 * t6w) <= 1.10 * H * W):
        return 6
    if max_m >= 4 and H % 4 == 0 and W % 4 == 0 and cin >= 128 and hw >= 64 and N * (H // 4) * (W // 4) >= 256:
        return 4
    if max_m >= 2 and H % 2 == 0 and W % 2 == 0 and cin >= 256 and hw >= 100 and N * (H // 2) * (W // 2) >= 1024:
        return 2
    # SMALL layers (16x16 ... 4x4 maps of the latent models, the inner levels of the 64^2-pixel model): measured in round 3 on the
    # bf16x3 pipe GEMM (tools/small_conv_bench.py -> profiles/r03_small_conv_bench.txt): F(2x2,3x3) beats the direct f32-MFMA kernel
    # 1.3-2.8x from 128 tiles up (split-K + 128-row tiles fill the chip: csrc/gemm_bf3p.hip), and F(4x4) whose 256-row GEMM tiles
    # would be mostly padding.  Needs whole 16-channel chunks (the pre-split operand layout).
    if small and max_m >= 2 and H % 2 == 0 and W % 2 == 0 and cin % 16 == 0 and cin >= 128 and hw >= 64 \
            and 128 <= N * (H // 2) * (W // 2) <= 8192:      # (narrow layers with more tiles: 1.0x, N32 64x64 128->128)
        return 2
    return 0


def winograd_wgrad_tile(N: int, H: int, W: int, cin: int, cout: int, max_m: int = 6, allow8: bool = False) -> int:
    """Tile m of the Winograd-domain WEIGHT gradient (csrc/winograd_wgrad.hip) for this 3x3 layer, or 0 = the direct
    kernel (conv_wgrad.hip).  The (m+2)^2 TN GEMMs contract over the tiles, so what matters is a long K (tiles) and
    operands wide enough for 128-wide MFMA tiles; ragged m = 6 tiles only pay while the edge waste stays small."""
    if cin % 4 or cout % 4 or cin < 64 or cout < 64:
        return 0
    # m = 8: where the forward takes it (the gradient contracts the transposed planes of V that forward kept: same tile both ways)
    if allow8 and max_m >= 8 and cin % 32 == 0 and _tile8_ok(N, H, W, cin, cout, 512 if allow8 is True else int(allow8)):
        return 8
    t6h, t6w = -(-H // 6), -(-W // 6)
    if max_m >= 6 and N * t6h * t6w >= 900 and (6 * t6h) * (6 * t6w) <= 1.10 * H * W:
        return 6
    if max_m >= 4 and H % 4 == 0 and W % 4 == 0 and N * (H // 4) * (W // 4) >= 256:
        return 4
    return 0


# --------------------------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------------------------
class UNetModel(nn.Module):
    """Drop-in for the reference ``UNetModel`` (openaimodel.py:416-759); see the module docstring."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 condition_key="concat"):
        super().__init__()
        # --- argument checks, same messages / conditions as openaimodel.py:474-491 ------------------------------
        if use_spatial_transformer:
            assert context_dim is not None, \
                'Fool!! You forgot to include the dimension of your cross-attention conditioning...'
        if context_dim is not None:
            assert use_spatial_transformer, \
                'Fool!! You forgot to use the spatial transformer for your cross-attention conditioning...'
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, 'Either num_heads or num_head_channels has to be set'
        if num_head_channels == -1:
            assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        # --- hot-path scope: the four reference templates + both ResBlock / resampling variants ------
        unsupported = []
        if dims != 2:
            unsupported.append(f"dims={dims}")
        if num_classes is not None:
            unsupported.append("num_classes")
        if n_embed is not None:
            unsupported.append("n_embed")
        if unsupported:
            raise NotImplementedError("bbdm_amd.UNetModel does not implement: " + ", ".join(unsupported))

        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32          # use_fp16 is a no-op in the reference too (openaimodel.py:25-29)
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = False
        self.condition_key = condition_key
        self.use_scale_shift_norm = use_scale_shift_norm
        self.use_new_attention_order = use_new_attention_order

        mc = model_channels
        ted = mc * 4
        self.time_embed = nn.Sequential(nn.Linear(mc, ted), nn.SiLU(), nn.Linear(ted, ted))

        def res(ch, out_ch, up=False, down=False):
            return ResBlock(ch, ted, dropout, out_channels=out_ch, use_scale_shift_norm=use_scale_shift_norm,
                            up=up, down=down)

        self.use_spatial_transformer = use_spatial_transformer
        self.context_dim = context_dim

        def attn(ch, heads):
            if not use_spatial_transformer:
                return AttentionBlock(ch, num_heads=heads, num_head_channels=num_head_channels,
                                      use_new_attention_order=use_new_attention_order)
            # openaimodel.py:546-565: heads = ch // num_head_channels when that is given (else num_heads); with legacy=True
            # the SpatialTransformer's head width is ch // heads
            nh = num_heads if num_head_channels == -1 else ch // num_head_channels
            d_head = ch // nh
            if d_head not in (16, 32, 64):
                raise NotImplementedError(f"bbdm_amd.UNetModel: SpatialTransformer head width {d_head} (channels {ch} / "
                                          f"{nh} heads) is not implemented; the attention kernel takes 16, 32 or 64")
            return SpatialTransformer(ch, nh, d_head, depth=transformer_depth, context_dim=context_dim)

        # construction order == openaimodel.py:518-691 (keeps RNG consumption, hence seeded init, identical)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, mc, 3, padding=1))])
        input_block_chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * mc)]
                ch = mult * mc
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, num_heads), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [res(ch + ich, mc * mult)]
                ch = mc * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown
                                  else Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), _zero_(nn.Conv2d(mc, out_channels, 3, padding=1)))

        self._plans: Dict[tuple, "_Plan"] = {}
        self.max_cached_plans = 4
        self._freqs: Optional[torch.Tensor] = None
        self.op_profile: Optional[list] = None      # set to a list to collect per-op HIP-event timings (bench.py)
        # ---- planner attributes (A/B runs and tests set them on the model before the first plan of a shape is built; plans are
        # cached per setting; nothing here reads the environment -- bench.py's --set ATTR=INT covers command-line A/Bs) ----
        # inference plans replay their launches as ONE hipGraph (None = yes on a GPU; False = launch by launch, e.g. under rocprofv3)
        self.hip_graph: Optional[bool] = None
        # 3x3 convolutions of wide layers through Winograd F(m x m, 3x3) (csrc/winograd.hip; `winograd_tile` picks per layer):
        # largest output tile allowed: 8 (default), 6, 4, 2, or 0 = direct kernel everywhere (bit-closer parity, A/B).  8 = F(8x8, 3x3) on
        # the large layers (>= 512 tiles, whole 128-channel blocks; inference plans, and training plans under ``winograd_train8``): -15 %
        # tile-GEMM work and transformed bytes for ~7x the fp32 rounding error of m = 6 -- round 5, bf16x3 planes: the C2 step 108.0 -> 97.1
        # ms, its parity against the reference 1.6e-5 -> 1.1e-4 of the 1e-3 bar (BASELINE.json north_star); round 6, fp16-pair planes
        # (``gemm_h2``): 77.8 ms at 9.7e-5, and 85.5 ms at 1.1e-5 with ``winograd = 6`` -- the one-line opt-out for a caller who wants
        # round 4's accuracy (the reference's own GPU path runs its convolutions in TF32 by default: ~1e-3)
        self.winograd: int = 8
        # fold GroupNorm -> FiLM -> SiLU (and an up-sampling ResBlock's nearest x2) into the Winograd input transform
        self.winograd_fuse_groupnorm: bool = True
        # small 3x3 layers (128 ... 8192 F(2x2) tiles) on F(2x2,3x3) + the bf16x3 GEMM instead of the direct f32-MFMA kernel
        self.winograd_small: bool = True
        # conv3x3(nearest x2 (x)) of inference plans as four phase filters on x itself (input transform and GEMM A operand 4x smaller)
        self.upsample_phases: bool = True
        # Winograd tile GEMMs and wide 1x1 layers on the BF16 matrix core with fp32 accuracy (three-way exact operand split, six
        # product terms; csrc/gemm_bf3*.hip) instead of the f32 MFMA, which gfx950 runs at 1/16 of the bf16 rate.  False: f32 MFMA
        # (bench.py's strict-f32 A/B)
        self.gemm_bf3: bool = True
        # ... with the A operand split into its three bf16 planes by the Winograd input transform (csrc/gemm_bf3p.hip: the GEMM's main
        # loop is LDS-DMA copies + MFMAs).  False: gemm_bf3.hip on fp32 V (tests: the two pipelines against each other)
        self.gemm_bf3p: bool = True
        # inference: the tile GEMMs of Winograd layers whose input is bounded by its GroupNorm coefficients (every 3x3 conv inside a
        # ResBlock) on TWO fp16 planes per operand under a provable power-of-two scale (csrc/h2_split.h; round 6): three MFMA terms
        # instead of six, 4 B per operand element instead of 6, and half the roundings of the fp32 accumulator -- faster AND closer to
        # the reference than the bf16x3 planes.  False: bf16x3 planes everywhere (round 5's plans; A/B)
        self.gemm_h2: bool = True
        # ... and training plans (0 = bf16x3 planes in every direction, round 5's plans): 1 = the FORWARD tile GEMMs, whose operand is
        # GroupNorm-bounded exactly as in sampling (the transposed copy of V the weight gradient contracts stays the exact bf16 split);
        # 2 = also the DATA-GRADIENT tile GEMMs, dY scaled by its measured maximum (one bbdm_absmax_rows_f32 pass per layer); 3 = also the
        # Winograd-domain WEIGHT gradient (the transposed copy of V and dM = A dY A^T as fp16 pairs, the TN GEMM on three terms).  A
        # gradient tensor has no a-priori range: elements more than ~2^17 below its maximum lose bits of their second plane (their
        # absolute error stays 2^-25 of the maximum), which the all-248-gradients tests bound at the benchmarked plan.
        self.gemm_h2_train: int = 3
        # ... and the wide 1x1 convolutions whose input carries a bound -- the skip projections of the ResBlocks (their raw input is bounded
        # by its own GroupNorm statistics: |x| <= sqrt(sum of squares) per group) and the qkv projections (a GroupNorm output) -- on
        # bbdm_conv1x1_h2q_f32 (inference plans).  False: bbdm_conv1x1_bf3q_f32 / bbdm_conv1x1_bf3_f32
        self.conv1x1_h2: bool = True
        # ... and the long-sequence attention (the pre-split pair bbdm_attention_kv_planes_f32 + bbdm_attention_planes_f32) on
        # bbdm_attention_*_h2_f32: q, k, v under the provable bound of the qkv projection (GroupNorm bound x max row L1 of its weight + max
        # |bias|), the softmax weights under their exact bound 1 (inference plans).  False: bf16x3 planes
        self.attn_h2: bool = True
        # 1x1 layers with fewer 256 x 128 output tiles than this leave the wide bf16x3 kernels for the small-problem kernel
        self.bf3_min_tiles: int = 256
        # ... csrc/gemm_bf3p.hip: gemm_bf3s_kernel (one launch, 64 channels per step); False: the split-K f32-MFMA kernel + reduction
        self.conv1x1_small: bool = True
        # inference plans: a Winograd layer of at most this many tiles forms its fused GroupNorm coefficients INSIDE its input
        # transform (bbdm_winograd_input_bf3p_gn_f32) instead of reading what a bbdm_groupnorm_coeffs_f32 launch wrote -- one launch
        # fewer per GroupNorm, the same bits (C5 3.25 -> 3.22 ms); above it every thread repeating the fp64 fold costs the HBM-bound
        # transforms more than the launch (round 4: +2.3 ms on the C2 step).  0: always the separate launch
        self.gn_in_transform: int = 1024
        # GroupNorm statistics accumulated by the kernel that produces the tensor (conv epilogue / Winograd output transform)
        # instead of a separate pass that re-reads it.  False: stand-alone bbdm_groupnorm_stats_f32 everywhere (tests)
        self.fuse_stats: bool = True
        # inference: Winograd layers whose tile GEMM is HBM-bound -- at most this many output channels and >= 512 MB of plane bytes, i.e.
        # the Cout = 128 layers of the 256^2 level (10 - 17 B of operands per 256 FLOP) -- keep V as fp32 rows (4 B per element instead
        # of the planes' 6; gemm_bf3.hip splits them under its idle matrix pipe): a third fewer operand bytes in both the input
        # transform and the GEMM (round 5: C2 -1.1 ms).  0 = planes everywhere
        self.fp32_v_max_cout: int = 128
        # inference: the four phase filters of an up-sampling conv as 2 x 2 filters, F(7x7, 2x2) on the 8-point transform, where the
        # layer would take F(6x6, 3x3) (see phase_filter_tile; round 5: C2 -1.7 ms); False = F(6x6, 3x3) on the zero-padded 3 x 3 filters
        self.upsample_f72: bool = True
        # training: set by dist_utils.accumulation_sync for the micro-steps whose parameter gradients no hook has to observe -- the
        # backward then adds them to the existing ``.grad`` tensors itself (bbdm_amd/autograd.py: _accumulate_in_place)
        self.grad_in_place: bool = False
        # training: weight gradients of the 3x3 layers in the Winograd domain (csrc/winograd_wgrad.hip), largest tile allowed;
        # 0: the direct kernel (conv_wgrad.hip) everywhere (tests)
        self.winograd_wgrad: int = 8
        # training plans and F(8x8, 3x3) (``winograd`` = 8, the bf16x3 pipeline on): 0 = never (round 5's plans: m <= 6 both ways), 1 = the
        # data-gradient convolutions only (they are forward convolutions of dY with the flipped filters: nothing new), 2 = also the forward
        # and the Winograd-domain weight gradient of the layers that keep V (dY transform A (8 -> 10 points), finish G^T . G in fp64).
        # C4: -31 % tile-GEMM work on the 32^2 maps (m = 4 before: 32 is no multiple of 6), -12 % on the 64^2 maps
        self.winograd_train8: int = 2
        # least number of 8x8 tiles for a layer to take F(8x8, 3x3): below it the 100 x 6 B of weight planes per weight are the bound
        self.winograd8_min_tiles: int = 512
        # inference plans: the 1x1 skip projection of a channel-changing ResBlock on a SECOND stream of the captured graph, beside the
        # block's first convolution, when its N H W Cin Cout lies in [min, max].  Measured at the end of round 5 (same box, ms per step,
        # one stream -> two): C3 14.89 -> 14.51 (projections of 0.06 - 0.33 ms next to tile GEMMs that leave rounds unfilled); C5 3.10 ->
        # 3.20 and C1 3.72 -> 3.84 (projections of 10 - 40 us: a fork and a join cost more than they hide); C2 96.9 / 97.3 -> 98.0 / 98.4 (its
        # kernels share one power budget: running two at once lowers the clock of both; with only its one projection inside the band,
        # 128 -> 512 at 128^2: 97.05 / 97.44 -> 97.58 / 97.55 -- hence the pixel cap).  The band admits C3's; max = 0: one stream
        self.side_stream_min_macs: int = 4_000_000_000
        self.side_stream_max_macs: int = 30_000_000_000
        self.side_stream_max_pixels: int = 131072
        # ... and training plans: the projection's forward launch, and in the gradient plan its weight gradient + data gradient (a
        # workspace of their own), forked at the top of the block's backward and joined before the GroupNorm backward that adds dX
        self.side_stream_train: bool = True
        # ... and the Winograd-domain weight gradients of the ResBlocks' 3x3 layers (dY transform -> TN GEMM -> finish, on the side
        # workspace) beside the data gradient of the same layer, from this many N H W Cin Cout up
        self.side_stream_wgrad: bool = True
        self.side_stream_wgrad_min_macs: int = 1_000_000_000
        # ... and, after an optimizer step, the re-packing of the data-gradient weight operands (not read before the backward) beside
        # the forward instead of at the start of the backward
        self.side_stream_dgrad_pack: bool = True

    # reference API kept as no-ops (openaimodel.py:703-719; convert_module_to_f16 is a stub there as well)
    def convert_to_fp16(self):
        pass

    def convert_to_fp32(self):
        pass

    # ----------------------------------------------------------------------------------------------------------
    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """openaimodel.py:721-759.  x: [N, C, H, W] fp32 on the GPU; returns [N, out_channels, H, W] fp32."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if self.dropout and self.training:
            # nn.Dropout is the identity in eval(): a checkpoint trained with dropout samples exactly; the masks of train()
            # mode are not implemented (the four reference templates set dropout 0.0)
            raise NotImplementedError(f"bbdm_amd.UNetModel: dropout={self.dropout} in train() mode is not implemented "
                                      "(sampling / eval() with such a config is supported)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import unet_apply          # training path (autograd.Function over the same kernels)
            return unet_apply(self, x, timesteps, context)
        return self.infer(x, timesteps, context)

    @torch.no_grad()
    def infer_step(self, x, step: int, context=None):
        """One sampling step's UNet call (BrownianBridgeModel.p_sample): every image at timestep ``step``.  Returns (the plan's own output
        buffer -- valid until the next call with this input shape --, the plan)."""
        x, ctx = self._check_inputs(x, context)
        plan = self._plan_for(x, training=False)
        return plan.run(x, int(step), ctx, None, True), plan

    def _check_inputs(self, x, context):
        _lib.require_gpu(x, context)
        if x.dtype != torch.float32:
            raise TypeError(f"bbdm_amd.UNetModel computes in fp32 like the reference; got {x.dtype}")
        ctx = None
        if self.condition_key != "nocond":
            if context is None:
                raise ValueError("context is required unless condition_key == 'nocond' (openaimodel.py:741-742)")
            ctx = context.contiguous().float()
        x = x.contiguous()
        cin = x.shape[1] + (ctx.shape[1] if ctx is not None else 0)
        if cin != self.in_channels:
            raise RuntimeError(f"expected {self.in_channels} input channels (x + context), got {cin}")
        return x, ctx

    @torch.no_grad()
    def infer(self, x, timesteps, context=None, out: Optional[torch.Tensor] = None, borrow: bool = False):
        """``borrow``: return the plan's own output buffer instead of a copy -- valid only until the next call with this input
        shape; for callers that consume it at once (BrownianBridgeModel.p_sample feeds it to the fused bridge kernel)."""
        x, ctx = self._check_inputs(x, context)
        plan = self._plan_for(x, training=False)
        t = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        return plan.run(x, t, ctx, out, borrow)

    def _plan_for(self, x, training: bool) -> "_Plan":
        N, _, H, W = x.shape
        key = (N, H, W, x.device.index, x.shape[1], training, self.winograd,
               self.winograd_fuse_groupnorm, self.gemm_bf3, self.gemm_bf3p, self.fuse_stats, self.winograd_wgrad, self.winograd_train8, self.winograd8_min_tiles, self.side_stream_min_macs, self.side_stream_max_macs, self.side_stream_max_pixels, self.side_stream_train, self.side_stream_wgrad, self.side_stream_wgrad_min_macs, self.bf3_min_tiles,
               self.winograd_small, self.upsample_phases, self.conv1x1_small, self.gn_in_transform,
               self.fp32_v_max_cout, self.upsample_f72, self.gemm_h2, self.gemm_h2_train, self.conv1x1_h2, self.attn_h2)
        plan = self._plans.pop(key, None)
        if plan is None:
            # a plan owns every activation (+ gradient twin when training) of its shape -- several GB at full size: keep the
            # few shapes a run alternates between (train batch, validation batch, the runner's 4-image sample), drop the rest
            while len(self._plans) >= self.max_cached_plans:
                self._plans.pop(next(iter(self._plans)))
            plan = _Plan(self, N, H, W, x.device, x.shape[1], training=training)
        self._plans[key] = plan                       # most recently used last
        return plan

    def invalidate_inputs(self):
        """Forget which caller tensors the cached plans already hold.

        A plan skips the copy of an input it has seen before -- the conditioning image of a sampling loop, the x_next the previous step
        wrote -- when the tensor object AND its autograd version counter are unchanged.  A write that bypasses the version counter (another
        HIP library through a raw pointer, a DLPack consumer, ``tensor.data_ptr()`` handed to a kernel) is invisible to that check: call
        this after such a write (or pass a fresh tensor) and the next forward copies its inputs again."""
        for plan in self._plans.values():
            plan._x_src = None
            plan._ctx_src = None

    def _apply(self, fn, *a, **k):
        # .to(device) / .cuda() / .float(): drop compiled plans, they hold device pointers
        self._plans = {}
        self._freqs = None
        return super()._apply(fn, *a, **k)

    def freqs(self, device):
        """util.py:160-163: evaluated on the CPU in fp32, then moved to the device."""
        if self._freqs is None or self._freqs.device != device:
            half = self.model_channels // 2
            f = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
            self._freqs = f.to(device)
        return self._freqs


# --------------------------------------------------------------------------------------------------------------
# shape-specialised execution plan
# --------------------------------------------------------------------------------------------------------------
class _Plan:
    """Static schedule of C-ABI calls for one (N, H, W): buffers are allocated once, every call has fixed arguments.

    Data layout in HBM: all activations fp32 NHWC.  Each ``th.cat([h, hs.pop()], 1)`` of the
    reference (openaimodel.py:752) is a pre-allocated buffer whose two channel slices are written directly by the
    producers of ``h`` and of the skip (pitch = total channels), so no concat copy exists.  Block-local
    temporaries (normalised activations, conv intermediates, qkv) are five role buffers shared by all blocks.
    """

    GROUPS = 32

    SAVED_ROLES = ("A", "H1", "A2", "XR", "QKV", "AT")     # activations the backward pass re-reads

    def _init_state(self, m, N: int, device, training: bool):
        """Emitter state shared by every plan built on these kernels (the UNet's, and bbdm_amd/first_stage_hip.py's)."""
        self.m, self.N, self.device, self.training = m, N, device, training
        self.tape: List[tuple] = []         # training: one record per layer, replayed in reverse by _emit_backward
        self.bops: List[tuple] = []         # training: backward op list
        self.ops: List[tuple] = []          # (fn_name, args with unresolved refs)
        self._side_ranges: List[tuple] = []         # (first op, end op, joining op) of launches that run on the plan's second stream
        self._side_stream = None
        self.convs: List[_PackedConv] = []
        self.bufs: List[_Buf] = []
        self._scratch: Dict[str, _Buf] = {}
        self._gn_count = 0
        self._params: List[nn.Parameter] = []
        self.lib = _lib.load()
        self._n_coeffs, self._coeff_need = 0, 1
        self._coeff_bufs = [(_LateTensor(), _LateTensor()), (_LateTensor(), _LateTensor())]
        self._ctx_tokens = None
        self._conv_ws_need = 0
        self._conv_ws = _LateTensor()             # split-K scratch shared by every conv of the plan
        self._conv_ws_floats = _LateInt()
        self._wino_v, self._wino_m = _LateTensor(), _LateTensor()      # Winograd V / M planes shared by every layer
        self._wino_v_need = self._wino_m_need = 0
        self._h2_layers: List[tuple] = []            # (gamma, beta, film offset or -1, C, zmax) per bounded GroupNorm output (_gn_bound)
        self._h2_bounds = _LateTensor()              # one float per entry, refreshed by ONE launch per forward (_launch_embedding)
        self._h2_table = None
        self._h2_dy_slots = 0                        # training: bound slots of gradient tensors (the tail of _h2_bounds, _dy_bound)
        self._h2_x_slots = 0                         # bound slots of raw block inputs (_stats_bound)
        self._saved_V: Dict[int, tuple] = {}         # training: id(conv weight) -> (V kept by the forward, tile m)
        self._fused_train = set()                    # training: id(conv weight) of layers whose GN->SiLU input was never materialised
        self.film, self.film_total, self.resblocks, self._film_key, self.film_wp = None, 0, [], None, None
        # (id(buffer), channel offset) -> record of the conv op that LAST wrote that channel slice: where a GroupNorm
        # consumer can ask the producer to accumulate its statistics (bbdm_*_stats_f32) instead of re-reading the tensor
        self._writers: Dict[tuple, dict] = {}
        self.fused_stats = 0

    def _allocate(self):
        f32 = dict(dtype=torch.float32, device=self.device)
        for pair in self._coeff_bufs:
            for lt in pair:
                lt.t = torch.empty(self._coeff_need, **f32)
        self._conv_ws.t = torch.empty(max(1, self._conv_ws_need), **f32)
        self._conv_ws_floats.v = self._conv_ws_need
        self._h2_bounds.t = torch.zeros(max(1, len(self._h2_layers)) + self._h2_x_slots + self._h2_dy_slots, **f32)
        self._wino_v.t = torch.empty(max(1, self._wino_v_need), **f32)
        self._wino_m.t = torch.empty(max(1, self._wino_m_need), **f32)
        for b in self.bufs:
            b.tensor = torch.empty(max(1, b.numel), **f32)
        # GroupNorm statistics: one exact integer-limb accumulator per GroupNorm (csrc/stats_acc.h: order-independent sums, so a step is
        # bitwise reproducible), all zeroed by ONE fill per forward
        self._stats_slot_bytes = int(self.lib.bbdm_groupnorm_stats_bytes(self.N, self.GROUPS))
        self.stats = torch.zeros(max(1, self._gn_count) * self._stats_slot_bytes // 8, dtype=torch.int64, device=self.device)
        self._param_key = None
        self._bound: List[tuple] = []
        self._graph, self._graph_key = None, None
        self.op_flops = [self._algorithmic_flops(name, args) for name, args in self.ops]

    def __init__(self, m: UNetModel, N: int, H: int, W: int, device, cx: int, training: bool = False):
        self._init_state(m, N, device, training)
        self.H, self.W, self.cx = H, W, cx
        mc = m.model_channels
        ted = 4 * mc
        f32 = dict(dtype=torch.float32, device=device)

        # ---- embedding path buffers -------------------------------------------------------------------------------
        # rows per bbdm_linear_f32 / bbdm_linear_bwd_f32 call: both stage [rows x In] (+4 pad) floats in the 160 KB LDS and
        # take <= 64 rows; In = 4 * model_channels, so wide models (model_channels >= 160) get fewer rows per call
        self.emb_rows = max(1, min(64, (160 * 1024) // (4 * (ted + 4))))
        self.t_buf = torch.zeros(N, dtype=torch.int64, device=device)
        self.e0 = torch.empty(N, mc, **f32)
        self.e1 = torch.empty(N, ted, **f32)
        self.emb = torch.empty(N, ted, **f32)
        self.resblocks = [mod for mod in m.modules() if isinstance(mod, ResBlock)]
        self.film_off = {}
        off = 0
        for rb in self.resblocks:
            self.film_off[id(rb)] = off
            off += rb.emb_layers[1].out_features
        self.film_total = off
        self.film = torch.empty(N, off, **f32)
        self.film_w = torch.empty(off, ted, **f32)        # concatenation of every emb_layers.1 weight / bias
        self.film_b = torch.empty(off, **f32)
        self._film_key = None
        # inference: the concatenated FiLM weight is also kept PACKED (csrc/embed.hip: linear_packed_kernel streams it sequentially,
        # 51 MB at HBM rate instead of 1 TB/s); repacked whenever the weights change.  Training plans (new weights every step) keep
        # bbdm_linear_f32.
        self.film_wp = None
        if off and not training and self.lib.bbdm_linear_packed_supported(min(N, 32), ted, off):
            self.film_wp = torch.empty(self.lib.bbdm_linear_packed_bytes(off, ted), dtype=torch.uint8, device=device)

        # ---- input / output --------------------------------------------------------------------------------------
        cin = m.in_channels
        cpad = _round4(cin)
        self.x_in = torch.empty(N, cx, H, W, **f32)
        self.ctx_in = torch.empty(N, cin - cx, H, W, **f32) if cin > cx else None
        self.out_nchw = torch.empty(N, m.out_channels, H, W, **f32)
        x0 = self._new(N, H, W, cpad)
        self._op("bbdm_nchw_to_nhwc_f32", _TensorRef(self.x_in), cx,
                 _TensorRef(self.ctx_in) if self.ctx_in is not None else None, cin - cx, x0, x0.ld, cpad, N, H, W)

        # ---- walk the graph (openaimodel.py:744-759) ---------------------------------------------------------------
        # pass 1: channels / resolution of every input-block output, to size the concat buffers of the output blocks
        in_shapes = []
        h_c, hh, ww = None, H, W
        for blk in m.input_blocks:
            for layer in blk:
                if isinstance(layer, nn.Conv2d):
                    h_c = layer.out_channels
                elif isinstance(layer, ResBlock):
                    h_c = layer.out_channels
                    if layer.down:
                        hh, ww = hh // 2, ww // 2
                elif isinstance(layer, Downsample):
                    hh, ww = hh // 2, ww // 2
            in_shapes.append((h_c, hh, ww))
        mid_c = in_shapes[-1][0]
        cat_bufs: List[_View] = []
        hc, hh, ww = mid_c, in_shapes[-1][1], in_shapes[-1][2]
        for blk, (sc, sh, sw) in zip(m.output_blocks, in_shapes[::-1]):
            if (sh, sw) != (hh, ww):
                raise RuntimeError(f"bbdm_amd: input {H}x{W} is not divisible by the UNet's down-sampling factor")
            cat_bufs.append(self._new(N, hh, ww, hc + sc))
            for layer in blk:
                if isinstance(layer, ResBlock):
                    hc = layer.out_channels
                    if layer.up:
                        hh, ww = hh * 2, ww * 2
                elif isinstance(layer, Upsample):
                    hc = layer.out_channels
                    hh, ww = hh * 2, ww * 2
        n_in = len(m.input_blocks)

        def skip_dest(i):       # the skip of input block i is the tail slice of cat buffer (n_in - 1 - i)
            cb = cat_bufs[n_in - 1 - i]
            sc = in_shapes[i][0]
            return _View(cb.buf, cb.C - sc, cb.ld, cb.N, cb.H, cb.W, sc)

        def h_dest(j, c):       # the h entering output block j is the head slice of cat buffer j
            cb = cat_bufs[j]
            return _View(cb.buf, 0, cb.ld, cb.N, cb.H, cb.W, c)

        # pass 2: emit ops
        h = x0
        for i, blk in enumerate(m.input_blocks):
            h = self._emit_block(blk, h, skip_dest(i))
        h = self._emit_block(m.middle_block, h, h_dest(0, mid_c))
        for j, blk in enumerate(m.output_blocks):
            cb = cat_bufs[j]
            assert h.buf is cb.buf and h.off == 0
            dest = None
            if j + 1 < len(m.output_blocks):
                oc = [l.out_channels for l in blk if isinstance(l, (ResBlock, Upsample))][-1]
                dest = h_dest(j + 1, oc)
            h = self._emit_block(blk, cb, dest)
        # head: GN -> SiLU -> conv3x3 -> NCHW  (openaimodel.py:687-691,759)
        head_stats = self._gn_count
        # the head's few output channels take the one-thread-per-pixel kernel (csrc/conv_igemm.hip: conv3x3_narrow_kernel),
        # an HBM-bound pass with idle ALUs: GroupNorm -> SiLU is folded into its patch staging
        narrow_head = m.out_channels <= 8 and N * H * W >= 4096
        a, pre = self._gn_input(h, m.out[0], None, silu=1, name="A", fuse_direct=narrow_head)
        pc = self._conv(m.out[2], a.C)
        self._op("bbdm_conv2d_nhwc_f32", a, a.ld, _TensorRef(pc.packed), self._pref(pc.bias), None, 0,
                 _TensorRef(self.out_nchw), 0, 1, None, 0, *pre, N, a.H, a.W, a.C, pc.cout, 3)
        if training:
            self.tape.append(("head", m.out, h, a, head_stats))
            self._emit_backward(x0)

        self._allocate()

    @staticmethod
    def _algorithmic_flops(name, args):
        """2*MACs of the contraction an op performs (conv / linear / attention matmuls only)."""
        if name == "bbdm_conv1x1_bf3_f32":
            pixels, cin_pad, cout = args[8:11]
            cin = args[2].t.cin_true if hasattr(args[2].t, "cin_true") else cin_pad
            return 2.0 * pixels * cout * cin
        if name == "bbdm_conv2d_nhwc_f32":
            N, H, W, cin_pad, cout, ks = args[15:21]
            cin = args[2].t.cin_true if hasattr(args[2].t, "cin_true") else cin_pad
            return 2.0 * N * H * W * cout * cin * ks * ks
        if name == "bbdm_winograd_gemm_f32":        # the (m+2)^2 GEMMs actually executed: 2 (m+2)^2 tiles Cin Cout
            wm, N, H, W, cin_pad, cout = args[0], *args[4:9]
            cin = args[2].t.cin_true if hasattr(args[2].t, "cin_true") else cin_pad
            return 2.0 * wino_planes(wm) * wino_tiles(wm, N, H, W) * cin * cout
        if name == "bbdm_attention_f32":
            N, T, heads, ch = args[5:9]
            return 2.0 * 2.0 * N * heads * T * T * ch
        # training (gradient plan)
        if name == "bbdm_conv_wgrad_f32":
            N, H, W, cin, cout, ks = args[8:14]
            return 2.0 * N * H * W * cout * cin * ks * ks
        if name == "bbdm_conv3x3_winograd_wgrad_f32":   # the (m+2)^2 TN GEMMs actually executed
            wm, (N, H, W, cin, cout) = args[0], args[8:13]
            return 2.0 * wino_planes(wm) * wino_tiles(wm, N, H, W) * cin * cout
        if name == "bbdm_gemm_tn_batched_f32":          # (the Winograd-domain weight gradient on a V kept by the forward)
            batch, K, M, Nn = args[7:11]
            return 2.0 * batch * K * M * Nn
        if name == "bbdm_attention_bwd_f32":            # S recomputed + dP, dV, dQ, dK: five T x T x ch products
            N, T, heads, ch = args[10:14]
            return 5.0 * 2.0 * N * heads * T * T * ch
        return 0.0

    def activation_bytes(self):
        return sum(b.tensor.numel() * 4 for b in self.bufs)

    # ---- buffer helpers ---------------------------------------------------------------------------------------
    def _new(self, N, H, W, C) -> _View:
        b = _Buf(N * H * W * C)
        self.bufs.append(b)
        return _View(b, 0, C, N, H, W, C)

    def _tmp(self, name, N, H, W, C) -> _View:
        """Block-local temporary: one buffer per role, sized for the largest user (blocks run sequentially).
        In a training plan the roles the backward pass re-reads get their own buffer per use instead."""
        if self.training and name in self.SAVED_ROLES:
            return self._new(N, H, W, C)
        b = self._scratch.get(name)
        if b is None:
            b = _Buf(0)
            self._scratch[name] = b
            self.bufs.append(b)
        b.numel = max(b.numel, N * H * W * C)
        self._writers = {k: v for k, v in self._writers.items() if k[0] != id(b)}     # a new tensor lives there now
        return _View(b, 0, C, N, H, W, C)

    class _StatsRef:
        __slots__ = ("plan", "slot")

        def __init__(self, plan, slot):
            self.plan, self.slot = plan, slot

        def resolve(self):
            p = self.plan
            return p.stats.data_ptr() + self.slot * p._stats_slot_bytes

    def _pref(self, p):
        if p is None:
            return None
        self._params.append(p)
        return _ParamRef(p)

    def _op(self, name, *args):
        rec = (name, list(args))          # a list: a later consumer may patch statistics targets into its producer
        self.ops.append(rec)
        return rec

    def _packed(self, cls, weight, *args, **kw):
        return cls(weight, *args, **kw)

    def _conv(self, mod, cin_pad) -> _PackedConv:
        pc = self._packed(_PackedConv, mod.weight, mod.bias, cin_pad)
        self.convs.append(pc)
        return pc

    # ---- op emitters --------------------------------------------------------------------------------------------
    def _gn_apply(self, x: _View, gn: Optional[nn.GroupNorm], film_off: Optional[int], silu: int, resample: int,
                  name: str) -> _View:
        N = self.N
        Ho = x.H // 2 if resample == 1 else (x.H * 2 if resample == 2 else x.H)
        Wo = x.W // 2 if resample == 1 else (x.W * 2 if resample == 2 else x.W)
        y = self._tmp(name, N, Ho, Wo, x.C)
        if gn is not None:
            ref = _Plan._StatsRef(self, self._gn_count)
            self._gn_count += 1
            self._emit_stats(x, ref)
            film = None if film_off is None else _TensorRef(self.film, 4 * film_off)
            self._op("bbdm_groupnorm_apply_f32", x, x.ld, ref, self._pref(gn.weight), self._pref(gn.bias),
                     film, self.film_total, y, y.ld, N, x.H, x.W, x.C, self.GROUPS, float(gn.eps), silu, resample)
        else:
            self._op("bbdm_groupnorm_apply_f32", x, x.ld, None, None, None, None, 0, y, y.ld, N, x.H, x.W,
                     x.C, 1, 0.0, 0, resample)
        return y

    NO_PRE = (None, None, 0, 0)

    def _note_writer(self, dest: _View, rec, first_stat_arg: Optional[int]):
        """``rec`` (an op record) now holds the final value of ``dest``; ``first_stat_arg`` = index of its (stats0, cpg0,
        coff0, stats1, cpg1, coff1) arguments, or None when that kernel cannot accumulate statistics."""
        for k in [k for k in self._writers if k[0] == id(dest.buf) and k[1] < dest.off + dest.C and
                  dest.off < k[1] + self._writers[k]["C"]]:
            del self._writers[k]                                   # overlapped older writers are history
        self._writers[(id(dest.buf), dest.off)] = dict(rec=rec, C=dest.C, arg=first_stat_arg, used=0,
                                                       shape=(dest.N, dest.H, dest.W))

    def _emit_stats(self, x: _View, ref):
        """GroupNorm statistics of ``x`` into slot ``ref``: patched into the producer(s) of ``x`` when they are conv kernels
        that can accumulate them (csrc/winograd.hip: StatArgs), else the stand-alone streaming pass."""
        cpg = x.C // self.GROUPS
        parts, pos = [], x.off
        if self.m.fuse_stats and x.C % self.GROUPS == 0 and cpg % 4 == 0:
            while pos < x.off + x.C:
                w = self._writers.get((id(x.buf), pos))
                if (w is None or w["arg"] is None or w["used"] >= 2 or pos + w["C"] > x.off + x.C
                        or w["shape"] != (x.N, x.H, x.W)):
                    parts = []
                    break
                parts.append((w, pos - x.off))
                pos += w["C"]
        if not parts:
            self._op("bbdm_groupnorm_stats_f32", x, x.ld, ref, self.N, x.H * x.W, x.C, self.GROUPS)
            return
        for w, coff in parts:
            a = w["rec"][1]
            i = w["arg"] + 3 * w["used"]
            a[i], a[i + 1], a[i + 2] = ref, cpg, coff
            w["used"] += 1
        self.fused_stats += 1

    class _H2Ref:
        """Address of a bound slot of the plan (a device float, csrc/h2_split.h).  The plan's bounds tensor holds, in this order, the
        GroupNorm bounds (``kind`` "gn": one launch per forward fills them all), the bounds taken from GroupNorm statistics ("x": raw
        block inputs, bbdm_h2_stats_bound_f32) and the measured maxima of gradient tensors ("dy", training: zeroed in backward_begin)."""
        __slots__ = ("plan", "k", "kind")

        def __init__(self, plan, k, kind="gn"):
            self.plan, self.k, self.kind = plan, k, kind

        def resolve(self):
            p = self.plan
            base = {"gn": 0, "x": max(1, len(p._h2_layers)), "dy": max(1, len(p._h2_layers)) + p._h2_x_slots}[self.kind]
            return p._h2_bounds.t.data_ptr() + 4 * (base + self.k)

    def _h2_on(self, level: int) -> bool:
        """fp16-pair planes for this plan's forward (``level`` 1) / data-gradient (2) tile GEMMs?"""
        m = self.m
        if not (getattr(m, "gemm_h2", False) and m.gemm_bf3 and m.gemm_bf3p):
            return False
        return (not self.training and level == 1) or (self.training and getattr(m, "gemm_h2_train", 0) >= level)

    def _dy_bound(self, dy: _View):
        """Bound slot for a gradient tensor: its exact maximum, measured by one pass (the slots are zeroed in backward_begin)."""
        ref = _Plan._H2Ref(self, self._h2_dy_slots, "dy")
        self._h2_dy_slots += 1
        self._bop("bbdm_absmax_rows_f32", dy, dy.ld, self.N * dy.H * dy.W, dy.C, ref)
        return ref

    def _conv1x1_h2_ok(self, pixels: int, cin: int, cout: int) -> bool:
        """Would a 1x1 convolution of this size with a bounded input run on bbdm_conv1x1_h2q_f32?  (inference, wide layers: the small
        problems keep the small-problem kernel)"""
        return bool(getattr(self.m, "conv1x1_h2", True) and self._h2_on(1) and cin % 16 == 0 and cout % 4 == 0
                    and (pixels // 256) * -(-cout // 128) >= self.m.bf3_min_tiles)

    def _conv1x1_h2s_ok(self, pixels: int, cin: int, cout: int) -> bool:
        """... or, below that size, on the small-problem form bbdm_conv1x1_h2s_f32 (bound by its weight stream: a third fewer bytes)?"""
        return bool(not self.training and getattr(self.m, "conv1x1_h2", True) and self._h2_on(1) and self.m.conv1x1_small
                    and not self._conv1x1_h2_ok(pixels, cin, cout) and cin % 64 == 0 and cout % 4 == 0
                    and not (self.lib.bbdm_gemm_bf3_supported(pixels, cin, cout) and (pixels // 256) * -(-cout // 128) >= self.m.bf3_min_tiles))

    def _stats_bound(self, slot: int):
        """Bound slot for the RAW tensor whose GroupNorm statistics are accumulator ``slot``: the largest root-sum-of-squares over its
        (image, group) cells (csrc/groupnorm.hip: h2_stats_bound_kernel).  Emitted where the statistics are complete."""
        ref = _Plan._H2Ref(self, self._h2_x_slots, "x")
        self._h2_x_slots += 1
        self._op("bbdm_h2_stats_bound_f32", _Plan._StatsRef(self, slot), self.N, self.GROUPS, ref)
        return ref

    def _gn_bound(self, x: _View, gn, film_off):
        """Bound slot for GroupNorm(x) [-> FiLM] [-> SiLU] [-> average pool / nearest x2]: |value| <= |gamma (1 + s)| sqrt(n_g - 1) +
        |beta (1 + s) + t| whatever x holds (a z-score over n_g values is at most sqrt(n_g - 1); SiLU, averaging and copying do not
        grow it).  Inference plans only; None when the fp16-pair planes are off."""
        if not self._h2_on(1) or gn is None:
            return None
        n_g = x.H * x.W * (x.C // self.GROUPS)
        self._h2_layers.append((gn.weight, gn.bias, -1 if film_off is None else int(film_off), x.C, math.sqrt(max(n_g - 1, 1))))
        self._pref(gn.weight), self._pref(gn.bias)          # (their addresses are part of the plan's binding key)
        return _Plan._H2Ref(self, len(self._h2_layers) - 1)

    def _gn_input(self, x: _View, gn, film_off, silu: int, name: str, consumer=None, fuse_direct: bool = False,
                  upsample: bool = False):
        """Input of a conv that follows GroupNorm [-> FiLM] [-> SiLU] at the same resolution.

        Inference plans do not materialise the normalised tensor: they emit the statistics + a tiny per-(image, channel)
        coefficient kernel and return (x itself, the fused-producer arguments of bbdm_conv2d_nhwc_f32).  Training plans
        keep the explicit apply pass (the backward re-reads its output for the weight gradient).  The producer is folded
        into the direct conv kernel only for the narrow head (``fuse_direct``; elsewhere it costs more MFMA stalls than the pass it
        removes) but always into the HBM-bound Winograd input transform of ``consumer``, where it is
        free."""
        up = 2 if upsample else 1         # (``upsample``: the consumer convolves the nearest x2 upsampling of the activated tensor)
        fuse = fuse_direct or (consumer is not None and self.m.winograd_fuse_groupnorm
                                                        and self._winograd_ok(consumer, up * x.H, up * x.W, x.C))
        assert not upsample or (fuse and not self.training)
        if self.training:
            # Training plans materialise the activated tensor for the weight gradient -- unless that gradient will contract the V
            # this layer's forward keeps (Winograd layer, same tile both ways: _emit_winograd / conv_bwd): then nothing
            # re-reads the activation and GN -> FiLM -> SiLU folds into the input transform exactly as in sampling.
            if not (self.m.winograd_fuse_groupnorm and self._train_keeps_V(consumer, x)):
                bref = self._gn_bound(x, gn, film_off) if (consumer is not None and self._winograd_ok(consumer, x.H, x.W, x.C)) else None
                return self._gn_apply(x, gn, film_off, silu=silu, resample=0, name=name), _Pre(self.NO_PRE, h2=bref)
            self._fused_train.add(id(consumer.weight))
        elif not fuse:
            return self._gn_apply(x, gn, film_off, silu=silu, resample=0, name=name), self.NO_PRE
        N = self.N
        bref = self._gn_bound(x, gn, film_off) if consumer is not None else None
        ref = _Plan._StatsRef(self, self._gn_count)
        self._gn_count += 1
        self._emit_stats(x, ref)
        film = None if film_off is None else _TensorRef(self.film, 4 * film_off)
        if self._gn_folds_into_transform(consumer, x, up, h2=bref is not None):
            # small problem: the consumer's input transform forms the coefficients from the statistics (one launch fewer)
            return x, _GnPre((ref, None, x.C, silu), (self._pref(gn.weight), self._pref(gn.bias), film, self.film_total,
                                                       x.H * x.W, self.GROUPS, float(gn.eps)), h2=bref)
        k = self._n_coeffs
        self._n_coeffs += 1
        self._coeff_need = max(self._coeff_need, N * x.C)
        sc = _TensorRef(self._coeff_bufs[k % 2][0])
        bi = _TensorRef(self._coeff_bufs[k % 2][1])
        # (One small launch per fused GroupNorm: 41 x ~6.6 us per forward.  Folding it into the launch that completes the statistics --
        # last-workgroup ticket + the fold there, coefficients formed by the consumer -- was built and measured in round 4 and LOST at
        # every size: profiles/r04_stats_tail_negative.md.)
        self._op("bbdm_groupnorm_coeffs_f32", ref, self._pref(gn.weight), self._pref(gn.bias), film, self.film_total, sc, bi,
                 x.C, N, x.H * x.W, x.C, self.GROUPS, float(gn.eps))
        return x, _Pre((sc, bi, x.C, silu), h2=bref)

    def _gn_folds_into_transform(self, consumer, x: _View, up: int, h2: bool = False) -> bool:
        """Will ``consumer`` (a 3x3 conv on GN(x), nearest-upsampled ``up`` x) run as a Winograd layer on the pre-split planes, small
        enough for its input transform to form the GroupNorm coefficients itself?  Mirrors the choices of :meth:`_emit_conv`."""
        m = self.m
        if self.training or consumer is None or not m.gn_in_transform or not (m.gemm_bf3 and m.gemm_bf3p):
            return False
        H, W, cout = up * x.H, up * x.W, consumer.weight.shape[0]
        wm = self._winograd_ok(consumer, H, W, x.C)
        if not wm or wm == 8 or x.C % 16 or (x.C // self.GROUPS) % 2 or consumer.weight.shape[1] != x.C:
            return False                      # (m = 8: large layers only, no coefficient-folding input transform)
        small = bool(m.gemm_bf3 and m.gemm_bf3p and m.winograd_small)
        cands = [(wm, H, W, cout)]
        if up == 2 and m.upsample_phases:         # conv3x3(nearest x2 (x)) may run as four phase filters on x itself
            wl = phase_filter_tile(self.N, x.H, x.W, x.C, 4 * cout, m.winograd, small, m.upsample_f72 and not self.training)
            if wl >= min(wm, 6):
                if wl == 7:
                    return False                  # (F(7x7, 2x2): large layers only, no coefficient-folding input transform)
                cands = [(wl, x.H, x.W, 4 * cout)]
        for w_, h_, ww_, co_ in cands:
            if self._use_bf3(w_, h_, ww_, x.C, co_, h2=h2) not in ("p", "h") or \
                    self.lib.bbdm_winograd_tiles(w_, self.N, h_, ww_) > m.gn_in_transform:
                return False
        return True

    def _train_keeps_V(self, consumer, x: _View) -> bool:
        """Will the training forward of conv ``consumer`` on ``x`` keep its transformed input for the weight gradient?"""
        if consumer is None or not self.m.winograd_wgrad:
            return False
        w = consumer.weight
        wm = self._winograd_ok(consumer, x.H, x.W, x.C)
        return bool(wm) and w.shape[1] == x.C and self._wgrad_tile(x.H, x.W, x.C, w.shape[0]) == wm

    def _winograd_ok(self, mod, H, W, cin_pad, flags=0) -> int:
        """Winograd output tile for this conv (0 = direct kernel)."""
        w = mod.weight
        if not self.m.winograd or w.dim() != 4 or w.shape[2] != 3 or (flags & ~6) != 0:
            return 0
        return winograd_tile(self.N, H, W, cin_pad, w.shape[0], self.m.winograd,
                             small=bool(self.m.gemm_bf3 and self.m.gemm_bf3p and self.m.winograd_small),
                             allow8=self._allow8(2))

    def _allow8(self, level: int) -> int:
        """F(8x8, 3x3) for this plan's forward / weight gradient (``level`` 2) or data gradient (1)?  Needs the pre-split bf16x3 pipeline.
        0 = no, else the least number of 8x8 tiles a layer needs for it (UNetModel.winograd8_min_tiles)."""
        m = self.m
        ok = bool(m.gemm_bf3) if not self.training else (bool(m.gemm_bf3 and m.gemm_bf3p) and m.winograd_train8 >= level)
        return int(m.winograd8_min_tiles) if ok else 0

    def _wgrad_tile(self, H, W, cin, cout) -> int:
        """Tile of the Winograd-domain weight gradient of a 3x3 layer (0 = direct kernel); 8 only where the forward takes 8 and keeps the
        transposed planes of V (csrc/winograd.hip: the dY transform of m = 8 exists for the bf16x3 GEMM only)."""
        m = self.m
        a8 = self._allow8(2)
        if a8:
            t8 = self.lib.bbdm_winograd_tiles(8, self.N, H, W)
            if not (m.winograd >= 8 and self.lib.bbdm_gemm_bf3p_tn_supported(t8, cin, cout) and self.lib.bbdm_gemm_bf3p_supported(t8, cin, cout)):
                a8 = 0
        return winograd_wgrad_tile(self.N, H, W, cin, cout, m.winograd_wgrad, allow8=a8)

    def _use_bf3(self, wm, H, W, cin_pad, cout, keeps_V=False, h2=False):
        """Tile GEMMs of this layer on the bf16x3 kernels (fp32-accurate)?  False = f32 MFMA, True = csrc/gemm_bf3.hip (fp32 V,
        split while staged), "p" = csrc/gemm_bf3p.hip (V written pre-split by the input transform; with ``keeps_V`` -- the training
        backward contracts this layer's V again -- only where the weight-gradient GEMM takes the transposed planes too).  (Round 3
        also measured V as fp32 row units split by the GEMM's own waves, 4 B per element instead of 6: the input transforms gain what
        the tile GEMMs lose, profiles/r03_bf3q_bench.txt; that mode is gone.)"""
        if not self.m.gemm_bf3:
            return False
        tiles = self.lib.bbdm_winograd_tiles(wm, self.N, H, W)
        if (h2 and self.lib.bbdm_gemm_bf3p_supported(tiles, cin_pad, cout)
                and (not keeps_V or self.lib.bbdm_gemm_bf3p_tn_supported(tiles, cin_pad, cout))):
            return "h"          # ``h2``: the layer's input carries a bound -- two fp16 planes per operand (csrc/h2_split.h)
        if (not self.training and cout <= self.m.fp32_v_max_cout and wino_planes(wm) * tiles * cin_pad * 6 >= (512 << 20)
                and self.lib.bbdm_gemm_bf3_supported(tiles, cin_pad, cout)):
            return True         # HBM-bound tile GEMM: fp32 V (see UNetModel.fp32_v_max_cout)
        if self.m.gemm_bf3p and self.lib.bbdm_gemm_bf3p_supported(tiles, cin_pad, cout) and \
                (not keeps_V or self.lib.bbdm_gemm_bf3p_tn_supported(tiles, cin_pad, cout)):
            return "p"          # (keeps_V: the weight gradient then contracts the TRANSPOSED planes the input transform also writes)
        return bool(self.lib.bbdm_gemm_bf3_supported(tiles, cin_pad, cout))

    def _keeps_V(self, wm, H, W, cin_pad, cin, cout, upsample, bwd) -> bool:
        """Training forward: does this Winograd layer keep its fp32 V for the Winograd-domain weight gradient?"""
        return bool(self.training and not bwd and not upsample and cin_pad == cin and self.m.winograd_wgrad
                    and self._wgrad_tile(H, W, cin_pad, cout) == wm)

    def _emit_winograd(self, x, cin_pad, pw, pre, upsample, H, W, residual, res_ld, dest, flags, bwd=False):
        """input transform -> 16 batched GEMMs -> output transform (csrc/winograd.hip).  ``pw.phases``: H, W are x's; the GEMMs produce the
        4 Cout phase channels and the output transform scatters them over the [2H, 2W] result (BBDM_CONV_OUT_PHASES)."""
        emit = self._bop if bwd else self._op
        N, cout_y, wm = self.N, dest.C, pw.m
        cout = 4 * cout_y if pw.phases else cout_y          # channels of M
        if pw.phases:
            assert not bwd and not upsample and residual is None and (dest.H, dest.W) == (2 * H, 2 * W)
            flags |= 8
        tiles = self.lib.bbdm_winograd_tiles(wm, N, H, W)
        h2 = pw.bf3 == "h"             # V as two fp16 planes under the bound pre.h2 (4 B per element)
        split = pw.bf3 == "p" or h2    # "p": V as three bf16 planes: 6 B per element of the (shared, float-typed) scratch buffer
        assert not h2 or getattr(pre, "h2", None) is not None
        vb = (pre.h2,) if h2 else ()
        gb = (pre.h2, _TensorRef(pw.ubound)) if h2 else ()
        self._wino_v_need = max(self._wino_v_need, wino_planes(wm) * tiles * cin_pad * (3 if pw.bf3 == "p" else 2) // 2)
        # small layers: split-K tile GEMMs, the partial sums M[z] are added by the output transform (csrc/gemm_bf3p.hip: fwd_splits)
        ksplit = int(self.lib.bbdm_winograd_gemm_bf3p_splits(wm, N, H, W, cin_pad, cout)) if (split and not pw.phases) else 1
        self._wino_m_need = max(self._wino_m_need, ksplit * wino_planes(wm) * tiles * cout)
        vbuf = self._wino_v
        keeps = self._keeps_V(wm, H, W, cin_pad, pw.cin, cout, upsample, bwd)
        if keeps and not split:
            # training: this layer's weight gradient contracts the SAME transformed input (csrc/winograd_wgrad.hip) -- keep V
            # in a buffer of its own instead of re-running the input transform in the backward pass (memory: (m+2)^2/m^2 x
            # the activation, ~8 GB over the LBBDM-f4 UNet at batch 32, of the 288 GB)
            b = _Buf(wino_planes(wm) * tiles * cin_pad)
            self.bufs.append(b)
            vbuf = _View(b, 0, cin_pad, 1, 1, 1, cin_pad)
            self._saved_V[id(pw.weight)] = (vbuf, wm)
        if keeps and split:
            # ... as bf16 planes: the forward GEMM reads the shared scratch copy, the weight gradient the TRANSPOSED copy (rows =
            # channels, contraction index = tiles) that the same input-transform launch writes -- 6 B per element kept
            vt_h2 = h2 and self._h2_on(3)         # the weight gradient's TN GEMM on the fp16 pair as well: the transposed copy likewise
            nbytes = (self.lib.bbdm_gemm_h2p_tn_at_bytes if vt_h2 else self.lib.bbdm_gemm_bf3p_tn_at_bytes)(wino_planes(wm), tiles, cin_pad)
            vt = _TensorRef(torch.empty(nbytes, dtype=torch.uint8, device=self.device))
            self._saved_V[id(pw.weight)] = (vt, wm, "tr", pre.h2) if vt_h2 else (vt, wm, "tr")
            emit(_OpName("bbdm_winograd_input_f32", "bbdm_winograd_input_h2p_tr2_f32" if vt_h2 else
                         "bbdm_winograd_input_h2p_tr_f32" if h2 else "bbdm_winograd_input_bf3p_tr_f32"),
                 wm, x, x.ld, vbuf, *(pre or self.NO_PRE), 0, N, H, W, cin_pad, vt, *vb)
        elif isinstance(pre, _GnPre):
            assert split and not bwd, "the coefficient-folding input transform exists for the pre-split planes only"
            emit(_OpName("bbdm_winograd_input_f32", "bbdm_winograd_input_h2p_gn_f32" if h2 else "bbdm_winograd_input_bf3p_gn_f32"),
                 wm, x, x.ld, vbuf, *pre, 1 if upsample else 0, N, H, W, cin_pad, *pre.tail, *vb)
        else:
            emit(_OpName("bbdm_winograd_input_f32", "bbdm_winograd_input_h2p_f32" if h2 else
                         "bbdm_winograd_input_bf3p_f32" if split else "bbdm_winograd_input_f32"),
                 wm, x, x.ld, vbuf, *(pre or self.NO_PRE), 1 if upsample else 0, N, H, W, cin_pad, *vb)
        if ksplit > 1:
            emit(_OpName("bbdm_winograd_gemm_f32", "bbdm_winograd_gemm_h2p_splitk_f32" if h2 else "bbdm_winograd_gemm_bf3p_splitk_f32"),
                 wm, vbuf, _TensorRef(pw.packed), self._wino_m, N, H, W, cin_pad, cout, ksplit, *gb)
        else:
            gemm = _OpName("bbdm_winograd_gemm_f32", "bbdm_winograd_gemm_h2p_f32" if h2 else
                           "bbdm_winograd_gemm_bf3p_f32" if split else
                           "bbdm_winograd_gemm_bf3_f32" if pw.bf3 else "bbdm_winograd_gemm_f32")
            emit(gemm, wm, vbuf, _TensorRef(pw.packed), self._wino_m, N, H, W, cin_pad, cout, *gb)
        ks_tail = (ksplit,) if ksplit > 1 else ()
        if bwd and ksplit == 1:
            emit("bbdm_winograd_output_f32", wm, self._wino_m, None, residual, res_ld, dest, dest.ld, flags, N, H, W, cout_y)
        elif bwd:
            emit(_OpName("bbdm_winograd_output_f32", "bbdm_winograd_output_splitk_stats_f32"), wm, self._wino_m, None, residual,
                 res_ld, dest, dest.ld, flags, N, H, W, cout_y, None, 0, 0, None, 0, 0, ksplit)
        else:
            rec = emit(_OpName("bbdm_winograd_output_f32", "bbdm_winograd_output_splitk_stats_f32" if ksplit > 1 else
                               "bbdm_winograd_output_stats_f32"), wm, self._wino_m,
                       self._pref(pw.bias) if pw.bias is not None else None, residual, res_ld, dest, dest.ld, flags,
                       N, H, W, cout_y, None, 0, 0, None, 0, 0, *ks_tail)
            self._note_writer(dest, rec, 12)

    def _emit_conv(self, x: _View, mod, residual, dest: _View, res_ld: Optional[int] = None, flags: int = 0,
                   pre=None, upsample: bool = False):
        """``residual`` is an NHWC view, or (with flags & 2) a per-image [N][res_ld] tensor reference, or (flags & 4, Winograd path only)
        an NHWC view at half the resolution that is added nearest-upsampled x2.
        ``upsample``: the convolved tensor is the nearest x2 upsampling of ``x`` (Winograd path only)."""
        cout = mod.weight.shape[0]
        assert dest.C == cout, (dest.C, cout)
        if res_ld is None:
            assert residual is None or residual.C == cout
            res_ld = residual.ld if residual is not None else 0
        H, W = (2 * x.H, 2 * x.W) if upsample else (x.H, x.W)
        h2 = getattr(pre, "h2", None) is not None and self._h2_on(1)
        wm = self._winograd_ok(mod, H, W, x.C, flags)
        if wm and upsample and self.m.upsample_phases and residual is None and flags == 0 and mod.weight.shape[1] == x.C:
            # conv3x3(nearest x2 (x)) = four phase filters on x (Cin -> 4 Cout): same GEMM work, the input transform and the GEMM's
            # A operand shrink 4x -- taken where x's own tile grid earns the same Winograd tile as the upsampled one
            wl = phase_filter_tile(self.N, x.H, x.W, x.C, 4 * cout, self.m.winograd,
                                   bool(self.m.gemm_bf3 and self.m.gemm_bf3p and self.m.winograd_small),
                                   self.m.upsample_f72 and not self.training)
            if wl == 7 and self._use_bf3(7, x.H, x.W, x.C, 4 * cout, h2=h2) not in ("p", "h"):
                wl = 6                            # (F(7x7, 2x2) exists on the pre-split planes only)
            if wl >= min(wm, 6):               # (m = 8 at the upsampled size does not beat the phase filters' 4x smaller input transform)
                pw = self._packed(_PackedWinograd, mod.weight, mod.bias, x.C, wl, bf3=self._use_bf3(wl, x.H, x.W, x.C, 4 * cout, h2=h2),
                                  phases=True)
                self.convs.append(pw)
                self._emit_winograd(x, x.C, pw, pre, False, x.H, x.W, None, 0, dest, 0)
                return
        if wm:
            pw = self._packed(_PackedWinograd, mod.weight, mod.bias, x.C, wm, bf3=self._use_bf3(
                wm, H, W, x.C, cout, keeps_V=self._keeps_V(wm, H, W, x.C, mod.weight.shape[1], cout, upsample, False), h2=h2))
            self.convs.append(pw)
            self._emit_winograd(x, x.C, pw, pre, upsample, H, W, residual, res_ld, dest, flags)
            return
        assert not upsample
        # (_gn_folds_into_transform mirrors the choices above; if the two ever diverge, the statistics reference a _GnPre carries in
        # place of the coefficients must not reach a kernel that reads it as ``pre_scale`` -- round-4 advisor finding)
        if isinstance(pre, _GnPre):      # (a real error, not an assert: under python -O a statistics pointer would be read as coefficients)
            raise RuntimeError("bbdm_amd: a coefficient-folding producer reached a layer that is not a Winograd layer on the pre-split planes")
        ks = mod.weight.shape[2] if mod.weight.dim() == 4 else 1
        pixels = self.N * x.H * x.W
        xb = getattr(pre, "h2", None)
        if (ks == 1 and xb is not None and (pre is None or pre[0] is None) and flags == 0 and self._conv1x1_h2_ok(pixels, x.C, cout)):
            # wide 1x1 convolution whose input carries a bound: the fp16-pair planes (csrc/gemm_bf3p.hip: gemm_bf3q_pipe_kernel<NP = 2>)
            pb = self._packed(_PackedConvH2q, mod.weight, mod.bias, x.C)
            self.convs.append(pb)
            rec = self._op(_OpName("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_h2q_f32"), x, x.ld, _TensorRef(pb.packed), self._pref(pb.bias),
                           residual, res_ld, dest, dest.ld, pixels, x.C, cout, xb, _TensorRef(pb.ubound))
            self._note_writer(dest, rec, None)
            return
        if (ks == 1 and self.m.gemm_bf3 and (pre is None or pre[0] is None) and flags == 0
                and self.lib.bbdm_gemm_bf3_supported(pixels, x.C, cout)
                and (pixels // 256) * -(-cout // 128) >= self.m.bf3_min_tiles):
            # wide 1x1 convolutions / Linears (skip connections, qkv / proj_out, transformer projections): the fp32-accurate
            # bf16x3 GEMM with bias + residual in its epilogue (csrc/gemm_bf3.hip); small problems keep the split-K f32 kernel
            # ... on the pipelined kernel (csrc/gemm_bf3p.hip: gemm_bf3q_pipe_kernel, 200 - 214 instead of 165 - 192 TFLOP/s) where
            # Cout fills 256-column tiles; its 128-column form loses to gemm_bf3.hip's 8-wave workgroups
            q = (-(-cout // 128) * 128) % 256 == 0
            pb = self._packed(_PackedConvBf3q if q else _PackedConvBf3, mod.weight, mod.bias, x.C)
            self.convs.append(pb)
            rec = self._op(_OpName("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_bf3q_f32") if q else "bbdm_conv1x1_bf3_f32", x, x.ld,
                           _TensorRef(pb.packed), self._pref(pb.bias), residual, res_ld, dest, dest.ld, pixels, x.C, cout)
            self._note_writer(dest, rec, None)
            return
        if (ks == 1 and xb is not None and (pre is None or pre[0] is None) and flags == 0 and self._conv1x1_h2s_ok(pixels, x.C, cout)):
            pb = self._packed(_PackedConvH2q, mod.weight, mod.bias, x.C)
            self.convs.append(pb)
            rec = self._op(_OpName("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_h2s_f32"), x, x.ld, _TensorRef(pb.packed), self._pref(pb.bias),
                           residual, res_ld, dest, dest.ld, pixels, x.C, cout, xb, _TensorRef(pb.ubound))
            self._note_writer(dest, rec, None)
            return
        if (ks == 1 and self.m.gemm_bf3 and self.m.conv1x1_small and (pre is None or pre[0] is None) and flags == 0
                and x.C % 64 == 0 and cout % 4 == 0):
            # small 1x1 convolutions / Linears: bound by the length of a workgroup's chain of K steps, not by the matrix pipe -- the
            # small-problem bf16x3 kernel (64 channels per step, one launch) instead of the split-K f32 kernel + its reduction pass
            pb = self._packed(_PackedConvBf3q, mod.weight, mod.bias, x.C)
            self.convs.append(pb)
            rec = self._op(_OpName("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_bf3s_f32"), x, x.ld, _TensorRef(pb.packed), self._pref(pb.bias),
                           residual, res_ld, dest, dest.ld, pixels, x.C, cout)
            self._note_writer(dest, rec, None)
            return
        pc = self._conv(mod, x.C)
        self._conv_ws_need = max(self._conv_ws_need,
                                 self.lib.bbdm_conv_splitk_workspace_floats(self.N, x.H, x.W, x.C, pc.cout, pc.ks))
        fusable = bool(self.lib.bbdm_conv_stats_fusable(self.N, x.H, x.W, x.C, pc.cout, pc.ks))
        rec = self._op(_OpName("bbdm_conv2d_nhwc_f32", "bbdm_conv2d_nhwc_stats_f32"), x, x.ld, _TensorRef(pc.packed),
                       self._pref(pc.bias), residual, res_ld, dest, dest.ld, flags, self._conv_ws, self._conv_ws_floats,
                       *(pre or self.NO_PRE), self.N, x.H, x.W, x.C, pc.cout, pc.ks, None, 0, 0, None, 0, 0)
        self._note_writer(dest, rec, 21 if fusable else None)

    def _side_band(self, pixels: int, cin: int, cout: int) -> bool:
        """Does a 1x1 skip projection of this size run on the plan's second stream (UNetModel.side_stream_*)?"""
        m = self.m
        if self.training and not m.side_stream_train:
            return False
        return m.side_stream_min_macs <= pixels * cin * cout <= m.side_stream_max_macs and pixels <= m.side_stream_max_pixels

    def _side_band3(self, pixels: int, cin: int, cout: int) -> bool:
        """Does the Winograd-domain weight gradient of a ResBlock's 3x3 layer of this size run on the gradient plan's second stream?"""
        m = self.m
        return bool(m.side_stream_train and m.side_stream_wgrad) and pixels * cin * cout >= m.side_stream_wgrad_min_macs \
            and pixels <= m.side_stream_max_pixels

    def _emit_res(self, rb: ResBlock, x: _View, dest: Optional[_View]) -> _View:
        """ResBlock._forward (openaimodel.py:258-278)."""
        N = self.N
        film = rb.use_scale_shift_norm
        rs = 2 if rb.up else (1 if rb.down else 0)
        s1 = self._gn_count
        # The 1x1 skip projection of a channel-changing block reads only x: small inference plans launch it first, on a SECOND stream of
        # the captured graph, and join before the launch that adds it (the out conv's epilogue) -- it then runs beside the in conv's
        # launches instead of between them (UNetModel.side_stream_min_macs / _max_macs; kernels that own no shared workspace only)
        side, out = None, None
        early_skip = isinstance(rb.skip_connection, nn.Conv2d) and rs == 0 and self._side_band(N * x.H * x.W, x.C, rb.out_channels)

        def emit_early_skip():
            # (after the block's first GroupNorm input has been emitted: the statistics of x are complete there, and the projection on
            # the fp16-pair planes takes its bound from them -- the bound launch stays on the main stream, before the fork)
            nonlocal side, out
            out = dest if dest is not None else self._new(N, x.H, x.W, rb.out_channels)
            xbound = self._stats_bound(s1) if (rb.skip_connection.weight.shape[2] == 1
                                               and (self._conv1x1_h2_ok(N * x.H * x.W, x.C, rb.out_channels)
                                                    or self._conv1x1_h2s_ok(N * x.H * x.W, x.C, rb.out_channels))) else None
            k0 = len(self.ops)
            self._emit_conv(x, rb.skip_connection, None, out, pre=_Pre(self.NO_PRE, h2=xbound) if xbound is not None else None)
            if all(str(n) == "bbdm_conv1x1_bf3_f32" for n, _ in self.ops[k0:]):
                side = (k0, len(self.ops))
        # Up-sampling block, inference, both 3x3 convs on the Winograd path at the upsampled size: nothing is resampled explicitly.
        # GN -> SiLU -> nearest x2 folds into the input transform of in_layers[2] (index shift, csrc/winograd.hip: UP), and the skip path
        # x_upd(x) is the out conv's residual read at [h/2][w/2] (BBDM_CONV_RES_UPSAMPLE): the two gn_apply passes that wrote and
        # re-read 4x-size tensors (1.65 ms per C2 step) disappear and the input transform reads a quarter of the bytes.
        fold_up = (rs == 2 and not self.training and self.m.winograd_fuse_groupnorm and film
                   and not isinstance(rb.skip_connection, nn.Conv2d)
                   and self._winograd_ok(rb.in_layers[2], 2 * x.H, 2 * x.W, x.C)
                   and self._winograd_ok(rb.out_layers[3], 2 * x.H, 2 * x.W, rb.out_channels))
        if rs == 0:
            a, pre1 = self._gn_input(x, rb.in_layers[0], None, silu=1, name="A", consumer=rb.in_layers[2])
            if early_skip:
                emit_early_skip()
        elif fold_up:
            a, pre1 = self._gn_input(x, rb.in_layers[0], None, silu=1, name="A", consumer=rb.in_layers[2], upsample=True)
        else:       # up / down blocks resample between the activation and the conv: explicit apply pass
            a, pre1 = self._gn_apply(x, rb.in_layers[0], None, silu=1, resample=rs, name="A"), None
            if self._winograd_ok(rb.in_layers[2], a.H, a.W, a.C):
                pre1 = _Pre(self.NO_PRE, h2=self._gn_bound(x, rb.in_layers[0], None))    # (pooled / copied values keep the bound)
        xr = x if (rs == 0 or fold_up) else self._gn_apply(x, None, None, 0, rs, name="XR")
        oh, ow = (2 * x.H, 2 * x.W) if fold_up else (a.H, a.W)
        h1 = self._tmp("H1", N, oh, ow, rb.out_channels)
        if fold_up:
            self._emit_conv(a, rb.in_layers[2], None, h1, pre=pre1, upsample=True)
        elif film:
            self._emit_conv(a, rb.in_layers[2], None, h1, pre=pre1)
        else:       # h = h + emb_out[..., None, None] (openaimodel.py:275): per-image row added in the conv epilogue
            self._emit_conv(a, rb.in_layers[2], _TensorRef(self.film, 4 * self.film_off[id(rb)]), h1,
                            res_ld=self.film_total, flags=2, pre=pre1)
        s2 = self._gn_count
        a2, pre2 = self._gn_input(h1, rb.out_layers[0], self.film_off[id(rb)] if film else None, silu=1, name="A2",
                                  consumer=rb.out_layers[3])
        if out is None:
            out = dest if dest is not None else self._new(N, oh, ow, rb.out_channels)
        if isinstance(rb.skip_connection, nn.Conv2d):
            if not early_skip:
                # the projection reads the block input itself (pooled / copied for an up / down block: no larger): bounded by the
                # statistics the block's first GroupNorm took of it (slot s1, complete by now)
                xbound = self._stats_bound(s1) if (rb.skip_connection.weight.dim() == 4 and rb.skip_connection.weight.shape[2] == 1
                                                   and (self._conv1x1_h2_ok(N * xr.H * xr.W, xr.C, rb.out_channels)
                                                        or self._conv1x1_h2s_ok(N * xr.H * xr.W, xr.C, rb.out_channels))) else None
                self._emit_conv(xr, rb.skip_connection, None, out, pre=_Pre(self.NO_PRE, h2=xbound) if xbound is not None else None)
            self._emit_conv(a2, rb.out_layers[3], out, out, pre=pre2)
            if side is not None:
                self._side_ranges.append((side[0], side[1], len(self.ops) - 1))      # (.., the launch that reads the projection)
        elif fold_up:
            self._emit_conv(a2, rb.out_layers[3], xr, out, pre=pre2, flags=4)        # residual = x at half resolution
        else:
            self._emit_conv(a2, rb.out_layers[3], xr, out, pre=pre2)
        if self.training:
            self.tape.append(("res", rb, x, a, xr, h1, a2, out, s1, s2, rs))
        return out

    def _emit_attn(self, ab: AttentionBlock, x: _View, dest: Optional[_View]) -> _View:
        """AttentionBlock._forward (openaimodel.py:321-327)."""
        N, T, C = self.N, x.H * x.W, x.C
        ch = C // ab.num_heads
        s0 = self._gn_count
        a, pre = self._gn_input(x, ab.norm, None, silu=0, name="A")
        order = 1 if ab.use_new_attention_order else 0
        nb = int(self.lib.bbdm_attention_kv_planes_bytes(N, T, ab.num_heads, ch))
        attn_h2 = bool(nb and not self.training and getattr(self.m, "attn_h2", True) and self._h2_on(1) and (pre is None or pre[0] is None))
        # proj_out reads the attention's output, a convex combination of value rows: |a| <= max |v| <= bound(qkv)
        proj_h2 = self._conv1x1_h2_ok(N * T, C, C) or self._conv1x1_h2s_ok(N * T, C, C)
        abound = qb = None
        if (pre is None or pre[0] is None) and (attn_h2 or proj_h2 or self._conv1x1_h2_ok(N * T, C, 3 * C) or self._conv1x1_h2s_ok(N * T, C, 3 * C)):
            abound = self._gn_bound(x, ab.norm, None)                         # (a materialised GroupNorm output: bounded by its coefficients)
            if abound is not None and (self._conv1x1_h2_ok(N * T, C, 3 * C) or self._conv1x1_h2s_ok(N * T, C, 3 * C)):
                pre = _Pre(self.NO_PRE, h2=abound)
        qkv = self._tmp("QKV", N, x.H, x.W, 3 * C)
        self._emit_conv(a, ab.qkv, None, qkv, pre=pre)
        at = self._tmp("AT", N, x.H, x.W, C)
        lse = None
        if self.training:
            lse = _TensorRef(torch.empty(N * ab.num_heads * T, dtype=torch.float32, device=self.device))
        if abound is not None and (attn_h2 or proj_h2):
            # bound(qkv) = bound(GroupNorm output) x max row L1 of the projection's weight + max |bias|: one thread per forward
            rg = _RowL1Gain(ab.qkv.weight, ab.qkv.bias)
            self.convs.append(rg)
            qb = _Plan._H2Ref(self, self._h2_x_slots, "x")
            self._h2_x_slots += 1
            self._op("bbdm_h2_affine_bound_f32", abound, _TensorRef(rg.gain), qb)
        if attn_h2 and qb is not None:
            # ... the long-sequence pair on fp16-pair planes under that bound (csrc/attention.hip, NP = 2)
            nb2 = int(self.lib.bbdm_attention_kv_planes_h2_bytes(N, T, ab.num_heads, ch))
            self._wino_v_need = max(self._wino_v_need, (nb2 + 3) // 4)
            self._op(_OpName("bbdm_attention_kv_planes_f32", "bbdm_attention_kv_planes_h2_f32"), qkv, qkv.ld, self._wino_v, nb2, N, T,
                     ab.num_heads, ch, order, qb)
            self._op(_OpName("bbdm_attention_f32", "bbdm_attention_planes_h2_f32"), qkv, qkv.ld, at, at.ld, lse, N, T, ab.num_heads, ch,
                     order, self._wino_v, qb)
        elif nb:    # long sequences: K / V split into their bf16 operand planes once per head (in the Winograd scratch, idle here) instead
                    # of by each of the T / 128 workgroups that walk them (csrc/attention.hip: attn_kv_planes_kernel)
            self._wino_v_need = max(self._wino_v_need, (nb + 3) // 4)
            self._op("bbdm_attention_kv_planes_f32", qkv, qkv.ld, self._wino_v, nb, N, T, ab.num_heads, ch, order)
            self._op(_OpName("bbdm_attention_f32", "bbdm_attention_planes_f32"), qkv, qkv.ld, at, at.ld, lse, N, T, ab.num_heads, ch,
                     order, self._wino_v)
        else:
            self._op("bbdm_attention_f32", qkv, qkv.ld, at, at.ld, lse, N, T, ab.num_heads, ch, order)
        out = dest if dest is not None else self._new(N, x.H, x.W, C)
        self._emit_conv(at, ab.proj_out, x, out, pre=_Pre(self.NO_PRE, h2=qb) if (qb is not None and proj_h2) else None)
        if self.training:
            self.tape.append(("attn", ab, x, a, qkv, at, lse, out, s0))
        return out

    def _context_tokens(self) -> Optional[_View]:
        """The cross-attention context as NHWC tokens [N, Hc*Wc, context channels padded to 4]: 'b c h w -> b (h w) c'
        (attention.py:175-176) of the tensor the UNet was called with; None = self-attention (condition_key 'nocond')."""
        if self.ctx_in is None:
            return None
        if self._ctx_tokens is None:
            N, Cc, Hc, Wc = self.ctx_in.shape
            v = self._new(N, Hc, Wc, _round4(Cc))
            self._op("bbdm_nchw_to_nhwc_f32", _TensorRef(self.ctx_in), Cc, None, 0, v, v.ld, v.C, N, Hc, Wc)
            self._ctx_tokens = v
        return self._ctx_tokens

    def _emit_transformer(self, st: "SpatialTransformer", x: _View, dest: Optional[_View]) -> _View:
        """SpatialTransformer.forward (attention.py:249-263).  NHWC activations ARE the token matrix [N*H*W, C]; every
        Linear is a 1x1 convolution on the matrix core, LayerNorm / GEGLU are streaming kernels (csrc/transformer.hip), the
        softmax(QK^T)V of both attentions is the streaming-softmax kernel with its own key/value source."""
        N, H, W = self.N, x.H, x.W
        heads, d = st.n_heads, st.d_head
        inner = heads * d
        train = self.training
        ctx = self._context_tokens()
        # training plans keep every intermediate the gradient plan re-reads in a buffer of its own (and the residual
        # stream out of place: LayerNorm's backward needs each block input); inference reuses one scratch buffer per role
        tmp = (lambda name, n, h, w, c: self._new(n, h, w, c)) if train else self._tmp
        s0 = self._gn_count
        a, pre = self._gn_input(x, st.norm, None, silu=0, name="A")
        hb = tmp("ST_H", N, H, W, inner)
        self._emit_conv(a, st.proj_in, None, hb, pre=pre)

        def layernorm(ln_mod, src: _View, name: str) -> _View:
            out = tmp(name, N, H, W, src.C)
            self._op("bbdm_layernorm_f32", src, src.ld, self._pref(ln_mod.weight), self._pref(ln_mod.bias), out, out.ld,
                     N * H * W, src.C, float(ln_mod.eps))
            return out

        def attention(att: "CrossAttention", xin: _View, kv_src: _View, res: _View):
            q = tmp("ST_QKV", N, H, W, inner)
            self._emit_conv(xin, att.to_q, None, q)
            kvb = tmp("ST_KV", kv_src.N, kv_src.H, kv_src.W, 2 * inner)
            kview = _View(kvb.buf, 0, kvb.ld, kv_src.N, kv_src.H, kv_src.W, inner)
            vview = _View(kvb.buf, inner, kvb.ld, kv_src.N, kv_src.H, kv_src.W, inner)
            self._emit_conv(kv_src, att.to_k, None, kview)
            self._emit_conv(kv_src, att.to_v, None, vview)
            at = tmp("ST_AT", N, H, W, inner)
            lse = _TensorRef(torch.empty(N * heads * H * W, dtype=torch.float32, device=self.device)) if train else None
            self._op("bbdm_cross_attention_f32", q, q.ld, kview, vview, kvb.ld, at, at.ld, lse, N, H * W,
                     kv_src.H * kv_src.W, heads, d)
            out = self._new(N, H, W, inner) if train else res      # inference: + x in place (the residual aliases the output)
            self._emit_conv(at, att.to_out[0], res, out)
            return out, (att, xin, kv_src, q, kview, vview, at, lse)

        blocks = []
        for blk in st.transformer_blocks:
            h_in = hb
            ln1 = layernorm(blk.norm1, h_in, "ST_LN")
            h1, rec1 = attention(blk.attn1, ln1, ln1, h_in)                      # self-attention
            ln2 = layernorm(blk.norm2, h1, "ST_LN")
            h2, rec2 = attention(blk.attn2, ln2, ctx if ctx is not None else ln2, h1)
            ln3 = layernorm(blk.norm3, h2, "ST_LN")
            ff = blk.ff
            if not isinstance(ff.net[0], GEGLU):
                raise NotImplementedError("bbdm_amd: FeedForward without GEGLU (gated_ff=False) is not implemented")
            inner_ff = ff.net[2].in_features
            pr = tmp("ST_FF", N, H, W, 2 * inner_ff)
            self._emit_conv(ln3, ff.net[0].proj, None, pr)
            gl = tmp("ST_GL", N, H, W, inner_ff)
            self._op("bbdm_geglu_f32", pr, pr.ld, gl, gl.ld, N * H * W, inner_ff)
            h3 = self._new(N, H, W, inner) if train else h2
            self._emit_conv(gl, ff.net[2], h2, h3)
            blocks.append((blk, h_in, rec1, h1, rec2, h2, ln3, pr, gl))
            hb = h3
        out = dest if dest is not None else self._new(N, H, W, x.C)
        self._emit_conv(hb, st.proj_out, x, out)
        if train:
            self.tape.append(("st", st, x, a, s0, blocks, hb, out, ctx))
        return out

    def _emit_block(self, blk, h: _View, dest: Optional[_View]) -> _View:
        layers = list(blk)
        for k, layer in enumerate(layers):
            d = dest if k == len(layers) - 1 else None
            if isinstance(layer, nn.Conv2d):
                out = d if d is not None else self._new(self.N, h.H, h.W, layer.out_channels)
                self._emit_conv(h, layer, None, out)
                if self.training:
                    self.tape.append(("stem", layer, h, out))
                h = out
            elif isinstance(layer, ResBlock):
                h = self._emit_res(layer, h, d)
            elif isinstance(layer, AttentionBlock):
                h = self._emit_attn(layer, h, d)
            elif isinstance(layer, SpatialTransformer):
                h = self._emit_transformer(layer, h, d)
            elif isinstance(layer, Downsample):
                h = self._emit_down(layer, h, d)
            elif isinstance(layer, Upsample):
                h = self._emit_up(layer, h, d)
            else:
                raise NotImplementedError(f"bbdm_amd: unsupported layer {type(layer).__name__}")
        return h

    def _emit_down(self, ds: Downsample, x: _View, dest: Optional[_View]) -> _View:
        """Downsample.forward (openaimodel.py:161-163): stride-2 3x3 conv = stride-1 conv + keep every 2nd pixel
        (4x the minimal FLOPs on 2 layers of a configuration no reference template uses), or a 2x2 average pool."""
        N = self.N
        out = dest if dest is not None else self._new(N, x.H // 2, x.W // 2, ds.out_channels)
        if ds.use_conv:
            full = self._tmp("DSF", N, x.H, x.W, ds.out_channels)
            self._emit_conv(x, ds.op, None, full)
            self._op("bbdm_groupnorm_apply_f32", full, full.ld, None, None, None, None, 0, out, out.ld, N, x.H, x.W,
                     full.C, 1, 0.0, 0, 3)
        else:
            self._op("bbdm_groupnorm_apply_f32", x, x.ld, None, None, None, None, 0, out, out.ld, N, x.H, x.W, x.C, 1,
                     0.0, 0, 1)
        if self.training:
            self.tape.append(("down", ds, x, out))
        return out

    def _emit_up(self, us: Upsample, x: _View, dest: Optional[_View]) -> _View:
        """Upsample.forward (openaimodel.py:111-121): nearest x2, then an optional 3x3 conv."""
        N = self.N
        out = dest if dest is not None else self._new(N, x.H * 2, x.W * 2, us.out_channels)
        if us.use_conv and not self.training and self._winograd_ok(us.conv, 2 * x.H, 2 * x.W, x.C):
            u = None                 # nearest x2 folded into the Winograd input transform: the 4x tensor never exists
            self._emit_conv(x, us.conv, None, out, upsample=True)
        elif us.use_conv:
            u = self._gn_apply(x, None, None, 0, 2, name="XR")
            self._emit_conv(u, us.conv, None, out)
        else:
            u = None
            self._op("bbdm_groupnorm_apply_f32", x, x.ld, None, None, None, None, 0, out, out.ld, N, x.H, x.W, x.C, 1,
                     0.0, 0, 2)
        if self.training:
            self.tape.append(("up", us, x, u, out))
        return out

    # ---- backward plan (training) ----------------------------------------------------------------------------------
    class _GradRef:
        """Pointer into the flat parameter-gradient buffer of the current backward call (+ element offset)."""
        __slots__ = ("plan", "off")

        def __init__(self, plan, off):
            self.plan, self.off = plan, off

        def resolve(self):
            return self.plan._flat_grad.data_ptr() + 4 * self.off

    def _bop(self, name, *args):
        rec = (name, list(args))
        self.bops.append(rec)
        return rec

    def _emit_backward(self, x0: _View):
        """Walk the tape in reverse and emit the gradient ops.

        Every persistent activation buffer gets a gradient twin of the same geometry.  A tensor consumed by two
        layers (block input = GN path + skip path is handled inside one kernel; an input-block output feeds the next
        block AND an output block through the concat) receives its first gradient by overwrite and later ones by
        accumulation -- the concat consumer always runs first in reverse order and writes the whole buffer.
        """
        m, N, lib, dev = self.m, self.N, self.lib, self.device
        G = self.GROUPS
        # parameter -> offset in the flat gradient buffer.  The FiLM projections (every ResBlock's emb_layers.1) come FIRST, weights then
        # biases, in the order of the concatenated [film_total x 4 mc] GEMM that computes their gradients (_backward_embedding): that GEMM
        # then writes the parameter gradients in place (round 3 copied 2 x 21 slices out of a scratch tensor per micro-step); the other
        # parameters follow in m.parameters() order.
        self.param_list = list(m.parameters())
        self.grad_off, off = {}, 0
        for rb in self.resblocks:
            self.grad_off[id(rb.emb_layers[1].weight)] = off
            off += rb.emb_layers[1].weight.numel()
        self._film_b_off = off
        for rb in self.resblocks:
            self.grad_off[id(rb.emb_layers[1].bias)] = off
            off += rb.emb_layers[1].bias.numel()
        for p in self.param_list:
            if id(p) not in self.grad_off:
                self.grad_off[id(p)] = off
                off += p.numel()
        self.grad_total = off
        self._flat_grad = None
        gref = lambda p: _Plan._GradRef(self, self.grad_off[id(p)])

        gbufs: Dict[int, _Buf] = {}
        written = set()

        def gview(v: _View) -> _View:
            b = gbufs.get(id(v.buf))
            if b is None:
                b = _Buf(v.buf.numel)
                gbufs[id(v.buf)] = b
                self.bufs.append(b)
            return _View(b, v.off, v.ld, v.N, v.H, v.W, v.C)

        def first_write(v: _View) -> int:
            """0 = overwrite (first gradient reaching this buffer), 1 = accumulate."""
            k = id(v.buf)
            acc = 1 if k in written else 0
            written.add(k)
            return acc

        ws_floats, ws_doubles, colsum_c, ws_side_floats = [1], [1], [1], [1]
        side_chains: List[tuple] = []       # second-stream launch ranges of the block being emitted (joined at its last GroupNorm backward)

        def sstat(slot):
            return _Plan._StatsRef(self, slot)

        def conv_bwd(mod, x_in: _View, dy: _View, need_dx: bool, dx_name: str, cin_true=None, side: bool = False,
                     side_chain: bool = False, dy_bound=None):
            """wgrad + bias grad (+ dgrad into a scratch view).  dy: gradient of the conv output (pitch >= Cout).  ``side``: a 1x1
            layer whose launches may run on the plan's second stream -- its weight gradient takes a workspace of its own;
            ``side_chain``: a 3x3 layer whose Winograd-domain weight gradient on the kept planes may (that chain only)."""
            w = mod.weight
            cout, cin = w.shape[0], w.shape[1]
            ks = w.shape[2] if w.dim() == 4 else 1
            dy_ref = [dy_bound]                      # the bound slot of dY (fp16-pair planes): measured once, shared by wgrad and dgrad
                                                     # (``dy_bound``: the caller measured it already -- a gradient two layers read)
            wsn = ws_side_floats if side else ws_floats
            wsn[0] = max(wsn[0], lib.bbdm_conv_wgrad_workspace_floats(N, x_in.H, x_in.W, x_in.C, cout, ks))
            if x_in.C == cin:
                dw_dst = gref(w)
            else:                                   # padded stem input: gradient of the padding channels is dropped
                t = torch.empty(cout, x_in.C, ks, ks, dtype=torch.float32, device=dev)
                self._padded_wgrads.append((w, t, len(self.bops)))     # (.., index of the op that fills t: its segment copies it out)
                dw_dst = _TensorRef(t)
            wgm = (self._wgrad_tile(x_in.H, x_in.W, x_in.C, cout)
                   if (m.winograd_wgrad and ks == 3 and w.dim() == 4 and x_in.C == cin) else 0)
            dbias = gref(mod.bias) if mod.bias is not None else None
            saved = self._saved_V.get(id(w)) if wgm else None
            if wgm == 8 and not (saved is not None and saved[1] == 8 and len(saved) > 2):
                # m = 8 exists on the kept transposed planes only: a layer that did not keep them re-transforms x at m <= 6
                wgm = winograd_wgrad_tile(N, x_in.H, x_in.W, x_in.C, cout, min(6, m.winograd_wgrad))
                saved = None
            if saved is not None and saved[1] == wgm and len(saved) > 2:
                # the forward kept the TRANSPOSED bf16 planes of V: dY transform (transposed planes of dM + the fp32 plane (1, 1)) ->
                # the bf16x3 GEMM with the tiles as its K loop -> finish (csrc/gemm_bf3p.hip: bbdm_gemm_bf3p_tn_f32)
                P, Tp = wino_planes(wgm), lib.bbdm_winograd_tiles(wgm, N, x_in.H, x_in.W)
                T = wino_tiles(wgm, N, x_in.H, x_in.W)
                splits = lib.bbdm_gemm_bf3p_tn_splits(P, Tp, x_in.C, cout)
                w_h2 = len(saved) > 3                    # the forward kept V^T as fp16 pairs: dM on the pair too, under dY's measured maximum
                if w_h2:
                    dy_ref[0] = dy_ref[0] or self._dy_bound(dy)      # (on the main stream, before a side chain forks)
                n_dmt = ((lib.bbdm_gemm_h2p_tn_bt_bytes if w_h2 else lib.bbdm_gemm_bf3p_tn_bt_bytes)(P, Tp, cout) + 3) // 4          # floats
                o_dm11 = n_dmt
                o_du = o_dm11 + Tp * cout
                o_acc = (o_du + splits * P * x_in.C * cout + 1) & ~1
                # ``side`` (a ResBlock's 3x3 layer, UNetModel.side_stream_wgrad): the three launches of this weight gradient read dY,
                # the kept planes and their own workspace only -- on the plan's second stream beside the data gradient of the same
                # layer (whose tile GEMM leaves its last round half empty); joined at the end of the block
                chain = side_chain and self._side_band3(N * x_in.H * x_in.W, x_in.C, cout)
                wsn = ws_side_floats if chain else ws_floats
                wsn[0] = max(wsn[0], o_acc + 8 * cout + 2)        # (+ the column sums' limb cells: 4 x 8 B per channel)
                wsf = self._ws_f_side if chain else self._ws_f
                dMt, dm11, dU = _TensorRef(wsf, 0), _TensorRef(wsf, 4 * o_dm11), _TensorRef(wsf, 4 * o_du)
                k_chain = len(self.bops)
                if w_h2:
                    self._bop(_OpName("bbdm_winograd_dy_transform_bf3p_f32", "bbdm_winograd_dy_transform_h2p_f32"), wgm, dy, dy.ld, dMt, dm11,
                              N, x_in.H, x_in.W, cout, dy_ref[0])
                    self._bop(_OpName("bbdm_gemm_bf3p_tn_f32", "bbdm_gemm_h2p_tn_f32"), saved[0], dMt, dU, P, Tp, x_in.C, cout,
                              saved[3], float(lib.bbdm_winograd_input_gain(wgm)), dy_ref[0], float(lib.bbdm_winograd_dy_gain(wgm)))
                else:
                    self._bop("bbdm_winograd_dy_transform_bf3p_f32", wgm, dy, dy.ld, dMt, dm11, N, x_in.H, x_in.W, cout)
                    self._bop("bbdm_gemm_bf3p_tn_f32", saved[0], dMt, dU, P, Tp, x_in.C, cout)
                if dbias is not None and cout % 4 == 0:
                    # ... + the bias gradient in the same launch: column sums of dM's plane (1, 1) = the tile sums of dY
                    self._bop("bbdm_winograd_wgrad_finish_bias_f32", wgm, dU, splits, dw_dst, x_in.C, cout, dm11, T, dbias)
                else:
                    self._bop("bbdm_winograd_wgrad_finish_f32", wgm, dU, splits, dw_dst, x_in.C, cout)
                    if dbias is not None:
                        self._bop("bbdm_colsum_f32", dm11, cout, _TensorRef(wsf, 4 * o_acc), dbias, T, cout)
                if chain:
                    side_chains.append((k_chain, len(self.bops)))
            elif saved is not None and saved[1] == wgm:
                # the forward kept this layer's V: dY transform -> TN GEMM -> finish (the stages bbdm_conv3x3_winograd_wgrad_f32 chains)
                P, Tp = wino_planes(wgm), lib.bbdm_winograd_tiles(wgm, N, x_in.H, x_in.W)
                T = wino_tiles(wgm, N, x_in.H, x_in.W)
                splits = lib.bbdm_gemm_tn_splits(P, T, x_in.C, cout)
                o_du = P * Tp * cout
                o_acc = (o_du + splits * P * x_in.C * cout + 1) & ~1
                ws_floats[0] = max(ws_floats[0], o_acc + 8 * cout + 2)        # (+ the column sums' limb cells: 4 x 8 B per channel)
                dM, dU = _TensorRef(self._ws_f, 0), _TensorRef(self._ws_f, 4 * o_du)
                self._bop("bbdm_winograd_dy_transform_f32", wgm, dy, dy.ld, dM, N, x_in.H, x_in.W, cout)
                self._bop("bbdm_gemm_tn_batched_f32", saved[0], x_in.C, Tp * x_in.C, dM, cout, Tp * cout, dU, P, T, x_in.C, cout)
                self._bop("bbdm_winograd_wgrad_finish_f32", wgm, dU, splits, dw_dst, x_in.C, cout)
                if dbias is not None:       # column sums of the plane dM_(1,1) = the tile sums of dY (csrc/winograd_wgrad.hip)
                    self._bop("bbdm_colsum_f32", _TensorRef(self._ws_f, 4 * (wgm + 3) * Tp * cout), cout,
                              _TensorRef(self._ws_f, 4 * o_acc), dbias, T, cout)
            elif id(w) in self._fused_train:
                raise RuntimeError("bbdm_amd: internal error: the activation of a fused-producer training layer was not kept")
            elif wgm:
                ws_floats[0] = max(ws_floats[0], lib.bbdm_winograd_wgrad_workspace_floats(wgm, N, x_in.H, x_in.W, x_in.C, cout))
                self._bop("bbdm_conv3x3_winograd_wgrad_f32", wgm, x_in, x_in.ld, dy, dy.ld, dw_dst, dbias, self._ws_f, N,
                          x_in.H, x_in.W, x_in.C, cout)
            else:
                self._bop("bbdm_conv_wgrad_f32", x_in, x_in.ld, dy, dy.ld, dw_dst, dbias, self._ws_f_side if side else self._ws_f,
                          self._ws_f_side_floats if side else self._ws_f_floats, N, x_in.H, x_in.W, x_in.C, cout, ks)
            if not need_dx:
                return None
            dx = self._tmp(dx_name, N, x_in.H, x_in.W, x_in.C)
            wm = (winograd_tile(N, x_in.H, x_in.W, dy.C, x_in.C, m.winograd,
                                small=bool(m.gemm_bf3 and m.gemm_bf3p and m.winograd_small), allow8=self._allow8(1))
                  if (m.winograd and ks == 3 and w.dim() == 4 and x_in.C == cin) else 0)
            if wm:
                mode = self._use_bf3(wm, x_in.H, x_in.W, dy.C, x_in.C, h2=self._h2_on(2) and dy.C % 4 == 0 and dy.ld % 4 == 0)
                pk = _PackedWinograd(w, None, dy.C, wm, dgrad=True, bf3=mode)
                self.dconvs.append(pk)
                # fp16-pair planes: dY under its measured maximum (UNetModel.gemm_h2_train = 2)
                pre_dy = _Pre(self.NO_PRE, h2=dy_ref[0] or self._dy_bound(dy)) if mode == "h" else None
                self._emit_winograd(dy, dy.C, pk, pre_dy, False, x_in.H, x_in.W, None, 0, dx, 0, bwd=True)
                return dx
            pixels = x_in.N * x_in.H * x_in.W
            if (ks == 1 and x_in.C == cin and self._h2_on(2) and getattr(m, "conv1x1_h2", True) and dy.C % 16 == 0 and dy.ld % 4 == 0
                    and x_in.C % 4 == 0 and (pixels // 256) * -(-x_in.C // 128) >= m.bf3_min_tiles):
                # wide 1x1 layers on the fp16 pair: dX = dY W with dY under its measured maximum (as the 3x3 layers' data gradient)
                pk = _PackedDgradBf3(w, dy.C, planes="h")
                self.dconvs.append(pk)
                dy_ref[0] = dy_ref[0] or self._dy_bound(dy)
                self._bop(_OpName("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_h2q_f32"), dy, dy.ld, _TensorRef(pk.packed), None, None, 0, dx,
                          dx.ld, pixels, dy.C, x_in.C, dy_ref[0], _TensorRef(pk.ubound))
                return dx
            if (ks == 1 and m.gemm_bf3 and x_in.C == cin and lib.bbdm_gemm_bf3_supported(pixels, dy.C, x_in.C)
                    and (pixels // 256) * -(-x_in.C // 128) >= m.bf3_min_tiles):
                q = (-(-x_in.C // 128) * 128) % 256 == 0
                pk = _PackedDgradBf3(w, dy.C, planes=q)  # wide 1x1 layers: dX = dY W on the bf16x3 GEMM, like their forward
                self.dconvs.append(pk)
                self._bop(_OpName("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_bf3q_f32") if q else "bbdm_conv1x1_bf3_f32", dy, dy.ld,
                          _TensorRef(pk.packed), None, None, 0, dx, dx.ld, pixels, dy.C, x_in.C)
                return dx
            pk = _PackedDgrad(w, dy.C)
            self.dconvs.append(pk)
            self._conv_ws_need = max(self._conv_ws_need,
                                     lib.bbdm_conv_splitk_workspace_floats(N, x_in.H, x_in.W, dy.C, x_in.C, ks))
            self._bop("bbdm_conv2d_nhwc_f32", dy, dy.ld, _TensorRef(pk.packed), None, None, 0, dx, dx.ld, 0,
                      self._conv_ws, self._conv_ws_floats, None, None, 0, 0, N, x_in.H, x_in.W, dy.C, x_in.C, ks)
            return dx

        def gn_bwd(gn, x: _View, slot, film_off, da: _View, dadd: Optional[_View], silu, rs, dx: _View, acc: int):
            ws_doubles[0] = max(ws_doubles[0], lib.bbdm_groupnorm_bwd_workspace_doubles(N, x.C, G))
            film = None if film_off is None else _TensorRef(self.film, 4 * film_off)
            dfilm = None if film_off is None else _TensorRef(self.dfilm, 4 * film_off)
            self._bop("bbdm_groupnorm_bwd_f32", x, x.ld, sstat(slot), self._pref(gn.weight), self._pref(gn.bias), film,
                      self.film_total, da, da.ld, dadd, dadd.ld if dadd is not None else 0, dx, dx.ld, acc,
                      gref(gn.weight), gref(gn.bias), dfilm, self.film_total, self._ws_d2, N, x.H, x.W, x.C, G,
                      float(gn.eps), silu, rs)

        f32 = dict(dtype=torch.float32, device=dev)
        self.dfilm = torch.zeros(N, self.film_total, **f32)
        self.dconvs: List[_PackedDgrad] = []
        self._padded_wgrads: List[tuple] = []
        self._ws_f = _LateTensor()
        self._ws_f_side, self._ws_f_side_floats = _LateTensor(), _LateInt()
        self._bside_ranges: List[tuple] = []        # (first op, end op, joining op) of gradient-plan launches on the second stream
        self._ws_f_floats = _LateInt()
        self._ws_d = _LateTensor()
        self._ws_d2 = _LateTensor()
        self.dout_nchw = torch.empty_like(self.out_nchw)
        self.need_input_grad = False
        self.dctx_tokens: Optional[_View] = None    # d context through the cross-attention keys / values (SpatialTransformer)

        rec_ends = []       # len(self.bops) after each tape record, in backward order
        for rec in reversed(self.tape):
            kind = rec[0]
            if kind == "head":
                _, seq, h, a, slot = rec
                cpad = _round4(m.out_channels)
                dy = self._tmp("DOUT", N, a.H, a.W, cpad)
                self._bop("bbdm_nchw_to_nhwc_f32", _TensorRef(self.dout_nchw), m.out_channels, None, 0, dy, dy.ld, cpad,
                          N, a.H, a.W)
                dyv = _View(dy.buf, 0, dy.ld, N, a.H, a.W, cpad)
                da = conv_bwd(seq[2], a, dyv, True, "DA")
                dh = gview(h)
                gn_bwd(seq[0], h, slot, None, da, None, 1, 0, dh, first_write(h))
            elif kind == "res":
                _, rb, x, a, xr, h1, a2, out, s1, s2, rs = rec
                dout = gview(out)
                bside = None
                # dOut is read by the skip projection's and by the out conv's gradients: its maximum is measured once, on the main stream
                dref = self._dy_bound(dout) if (self._h2_on(2) and dout.C % 4 == 0 and dout.ld % 4 == 0) else None
                if isinstance(rb.skip_connection, nn.Conv2d):
                    # the projection's gradients read only dOut (complete before this block's backward starts) and the block input: on the
                    # second stream beside the block's own chain; joined before the launch that adds dXr (the last GroupNorm backward)
                    # (only where the data gradient takes the bf16x3 GEMM -- conv_bwd's rule: the direct kernel owns a shared workspace)
                    px = N * xr.H * xr.W
                    sb = (rs == 0 and self._side_band(px, xr.C, rb.out_channels) and bool(m.gemm_bf3)
                          and rb.skip_connection.weight.shape[1] == xr.C and bool(lib.bbdm_gemm_bf3_supported(px, dout.C, xr.C))
                          and (px // 256) * -(-xr.C // 128) >= m.bf3_min_tiles)
                    k0 = len(self.bops)
                    dxr = conv_bwd(rb.skip_connection, xr, dout, True, "DXR", side=sb, dy_bound=dref)
                    # (checked, not assumed -- and not an assert, which python -O drops: only the workspace-free pair may leave the main
                    # stream; if conv_bwd ever chooses other kernels than the rule above predicts, the launches simply stay in order)
                    if sb and [str(n) for n, _ in self.bops[k0:]] == ["bbdm_conv_wgrad_f32", "bbdm_conv1x1_bf3_f32"]:
                        bside = (k0, len(self.bops))
                else:
                    dxr = dout
                side_chains.clear()
                da2 = conv_bwd(rb.out_layers[3], a2, dout, True, "DA2", side_chain=True, dy_bound=dref)
                dh1 = self._tmp("DH1", N, h1.H, h1.W, h1.C)
                if rb.use_scale_shift_norm:
                    gn_bwd(rb.out_layers[0], h1, s2, self.film_off[id(rb)], da2, None, 1, 0, dh1, 0)
                else:           # d emb_out[n, c] = sum_hw d(h + emb_out)
                    gn_bwd(rb.out_layers[0], h1, s2, None, da2, None, 1, 0, dh1, 0)
                    colsum_c[0] = max(colsum_c[0], 4 * N * h1.C)        # limb cells (csrc/stats_acc.h): 4 words per sum
                    self._bop("bbdm_colsum_batched_f32", dh1, dh1.ld, self._ws_d,
                              _TensorRef(self.dfilm, 4 * self.film_off[id(rb)]), self.film_total, N, h1.H * h1.W, h1.C)
                da = conv_bwd(rb.in_layers[2], a, dh1, True, "DA", side_chain=True)
                dx = gview(x)
                for k0, k1 in ([bside] if bside is not None else []) + side_chains:
                    self._bside_ranges.append((k0, k1, len(self.bops)))
                side_chains.clear()
                gn_bwd(rb.in_layers[0], x, s1, None, da, dxr, 1, rs, dx, first_write(x))
            elif kind == "attn":
                _, ab, x, a, qkv, at, lse, out, s0 = rec
                T, C = x.H * x.W, x.C
                dout = gview(out)
                dat = conv_bwd(ab.proj_out, at, dout, True, "DAT")
                dqkv = self._tmp("DQKV", N, x.H, x.W, 3 * C)
                dwork = _TensorRef(torch.empty(N * ab.num_heads * T, **f32))
                self._bop("bbdm_attention_bwd_f32", qkv, qkv.ld, at, at.ld, dat, dat.ld, lse, dwork, dqkv, dqkv.ld, N, T,
                          ab.num_heads, C // ab.num_heads, 1 if ab.use_new_attention_order else 0)
                da = conv_bwd(ab.qkv, a, dqkv, True, "DA")
                dx = gview(x)
                gn_bwd(ab.norm, x, s0, None, da, dout, 0, 0, dx, first_write(x))
            elif kind == "st":
                # SpatialTransformer (attention.py:249-263) in reverse; BasicTransformerBlock._forward (attention.py:215-218):
                # x = attn1(norm1(x)) + x;  x = attn2(norm2(x), context) + x;  x = ff(norm3(x)) + x
                _, st, x, a, s0, blocks, h_last, out, ctx = rec
                heads, dh_ch = st.n_heads, st.d_head
                inner = heads * dh_ch
                H, W = x.H, x.W
                rows = N * H * W
                dout = gview(out)
                ring = ["ST_DHA", "ST_DHB", "ST_DHC"]                      # gradient of the residual stream: three live at most
                turn = [0]

                def next_dh():
                    turn[0] = (turn[0] + 1) % 3
                    return self._tmp(ring[turn[0]], N, H, W, inner)

                def accumulate(src: _View, dst: _View, acc: int):
                    self._bop("bbdm_groupnorm_bwd_f32", None, 0, None, None, None, None, 0, None, 0, src, src.ld, dst, dst.ld,
                              acc, None, None, None, 0, None, src.N, src.H, src.W, src.C, 1, 0.0, 0, 0)

                def ln_bwd(ln_mod, xin: _View, dy: _View, dadd: _View) -> _View:
                    ws_doubles[0] = max(ws_doubles[0], 8 * xin.C)      # [2][C] limb cells of 4 words
                    dxv = next_dh()
                    self._bop("bbdm_layernorm_bwd_f32", xin, xin.ld, self._pref(ln_mod.weight), dy, dy.ld, dadd, dadd.ld, dxv,
                              dxv.ld, gref(ln_mod.weight), gref(ln_mod.bias), self._ws_d2, rows, xin.C, float(ln_mod.eps))
                    return dxv

                def attn_bwd(arec, dres: _View) -> _View:
                    """dres = gradient of (to_out(attention) + residual); returns the gradient of the LayerNorm output."""
                    att, xin, kv_src, q, kview, vview, at, lse = arec
                    Tk = kv_src.H * kv_src.W
                    dat = conv_bwd(att.to_out[0], at, dres, True, "ST_DAT")
                    dq = self._tmp("ST_DQ", N, H, W, inner)
                    dkv = self._tmp("ST_DKV", kv_src.N, kv_src.H, kv_src.W, 2 * inner)
                    dk = _View(dkv.buf, 0, dkv.ld, kv_src.N, kv_src.H, kv_src.W, inner)
                    dv = _View(dkv.buf, inner, dkv.ld, kv_src.N, kv_src.H, kv_src.W, inner)
                    dwork = _TensorRef(torch.empty(N * heads * H * W, **f32))
                    self._bop("bbdm_cross_attention_bwd_f32", q, q.ld, kview, vview, kview.ld, at, at.ld, dat, dat.ld, lse, dwork,
                              dq, dq.ld, dk, dv, dkv.ld, N, H * W, Tk, heads, dh_ch)
                    dln = conv_bwd(att.to_q, xin, dq, True, "ST_DLN")
                    dsk = conv_bwd(att.to_k, kv_src, dk, True, "ST_DSK")
                    dsv = conv_bwd(att.to_v, kv_src, dv, True, "ST_DSV")
                    if kv_src is ctx:                       # context tokens: their gradient leaves through the UNet's d input
                        if self.dctx_tokens is None:
                            self.dctx_tokens = self._new(ctx.N, ctx.H, ctx.W, ctx.C)
                            accumulate(dsk, self.dctx_tokens, 0)
                        else:
                            accumulate(dsk, self.dctx_tokens, 1)
                        accumulate(dsv, self.dctx_tokens, 1)
                    else:                                   # self-attention: keys and values come from the same LayerNorm output
                        accumulate(dsk, dln, 1)
                        accumulate(dsv, dln, 1)
                    return dln

                dh = conv_bwd(st.proj_out, h_last, dout, True, ring[0])
                for blk, h_in, rec1, h1, rec2, h2, ln3, pr, gl in reversed(blocks):
                    ff = blk.ff
                    dgl = conv_bwd(ff.net[2], gl, dh, True, "ST_DGL")
                    dpr = self._tmp("ST_DPR", N, H, W, pr.C)
                    self._bop("bbdm_geglu_bwd_f32", pr, pr.ld, dgl, dgl.ld, dpr, dpr.ld, rows, gl.C)
                    dln3 = conv_bwd(ff.net[0].proj, ln3, dpr, True, "ST_DLN")
                    dh2 = ln_bwd(blk.norm3, h2, dln3, dh)
                    dln2 = attn_bwd(rec2, dh2)
                    dh1 = ln_bwd(blk.norm2, h1, dln2, dh2)
                    dln1 = attn_bwd(rec1, dh1)
                    dh = ln_bwd(blk.norm1, h_in, dln1, dh1)
                da = conv_bwd(st.proj_in, a, dh, True, "DA")
                dx = gview(x)
                gn_bwd(st.norm, x, s0, None, da, dout, 0, 0, dx, first_write(x))
            elif kind == "down":
                _, ds, x, out = rec
                dout = gview(out)
                dx = gview(x)
                if ds.use_conv:
                    dfull = self._tmp("DDSF", N, x.H, x.W, ds.out_channels)
                    self._bop("bbdm_groupnorm_bwd_f32", None, 0, None, None, None, None, 0, None, 0, dout, dout.ld, dfull,
                              dfull.ld, 0, None, None, None, 0, None, N, x.H, x.W, dfull.C, 1, 0.0, 0, 3)
                    dxs = conv_bwd(ds.op, x, dfull, True, "DA")
                    self._bop("bbdm_groupnorm_bwd_f32", None, 0, None, None, None, None, 0, None, 0, dxs, dxs.ld, dx, dx.ld,
                              first_write(x), None, None, None, 0, None, N, x.H, x.W, x.C, 1, 0.0, 0, 0)
                else:
                    self._bop("bbdm_groupnorm_bwd_f32", None, 0, None, None, None, None, 0, None, 0, dout, dout.ld, dx, dx.ld,
                              first_write(x), None, None, None, 0, None, N, x.H, x.W, x.C, 1, 0.0, 0, 1)
            elif kind == "up":
                _, us, x, u, out = rec
                dout = gview(out)
                dx = gview(x)
                du = conv_bwd(us.conv, u, dout, True, "DA") if us.use_conv else dout
                self._bop("bbdm_groupnorm_bwd_f32", None, 0, None, None, None, None, 0, None, 0, du, du.ld, dx, dx.ld,
                          first_write(x), None, None, None, 0, None, N, x.H, x.W, x.C, 1, 0.0, 0, 2)
            elif kind == "stem":
                _, conv, x, out = rec
                dout = gview(out)
                conv_bwd(conv, x, dout, False, "DX0")
                # d input (only when x / context require grad, e.g. a trainable SpatialRescaler context): own op list
                main, self.bops = self.bops, []
                pk = _PackedDgrad(conv.weight, dout.C)
                self.dconvs.append(pk)
                self.dx0 = self._tmp("DX0", N, x.H, x.W, x.C)
                self._bop("bbdm_conv2d_nhwc_f32", dout, dout.ld, _TensorRef(pk.packed), None, None, 0, self.dx0,
                          self.dx0.ld, 0, None, 0, None, None, 0, 0, N, x.H, x.W, dout.C, x.C, 3)
                self.bops_x0, self.bops = self.bops, main
            rec_ends.append(len(self.bops))
        self._segment_backward(rec_ends)
        self._ws_f.t = torch.empty(ws_floats[0], **f32)
        self._ws_f_floats.v = ws_floats[0]
        self._ws_f_side.t = torch.empty(ws_side_floats[0], **f32)
        self._ws_f_side_floats.v = ws_side_floats[0]
        self._ws_d.t = torch.empty(colsum_c[0], dtype=torch.float64, device=dev)
        self._ws_d2.t = torch.empty(ws_doubles[0], dtype=torch.float64, device=dev)
        # embedding-path backward scratch
        ted = 4 * m.model_channels
        self.d_emb = torch.empty(N, ted, **f32)
        self.d_e1 = torch.empty(N, ted, **f32)
        self._lin_ws = torch.empty(max(lib.bbdm_linear_bwd_workspace_floats(min(N, self.emb_rows), ted, self.film_total),
                                       lib.bbdm_linear_bwd_workspace_floats(min(N, self.emb_rows), ted, ted), 1), **f32)
        self._bbound: List[tuple] = []

    # Backward segments.  The gradient plan is cut into a few runs of whole layers (head side first) so that the autograd
    # graph can be a CHAIN of nodes, one per segment (bbdm_amd/autograd.py): the engine hands the parameter gradients of a
    # finished segment to their AccumulateGrad nodes -- and with them to DDP's reducer hooks (runners/BaseRunner.py:76) --
    # before it runs the next segment, so the bucketed RCCL all-reduce of the late layers overlaps the backward kernels of
    # the early ones instead of starting after the whole UNet's backward has been enqueued.
    N_SEGMENTS = 4

    @staticmethod
    def _record_params(rec):
        kind = rec[0]
        mods = []
        if kind == "head":
            mods = [rec[1][0], rec[1][2]]
        elif kind == "res":
            rb = rec[1]
            mods = [rb.in_layers[0], rb.in_layers[2], rb.out_layers[0], rb.out_layers[3]]
            if isinstance(rb.skip_connection, nn.Conv2d):
                mods.append(rb.skip_connection)
        elif kind == "attn":
            ab = rec[1]
            mods = [ab.norm, ab.qkv, ab.proj_out]
        elif kind == "st":
            mods = [rec[1]]
        elif kind == "down":
            mods = [rec[1].op] if rec[1].use_conv else []
        elif kind == "up":
            mods = [rec[1].conv] if rec[1].use_conv else []
        elif kind == "stem":
            mods = [rec[1]]
        return [p for mod in mods for p in mod.parameters()]

    def _segment_backward(self, rec_ends):
        """Cut the backward op list at layer boundaries into <= N_SEGMENTS runs of roughly equal parameter count.
        ``self.bsegs[k]`` = (first op, last op + 1, parameters whose gradients are complete when the run ends); k = 0 is
        the head side (runs first).  The embedding path (time_embed + every ResBlock's emb_layers: their gradients need the
        d FiLM rows of ALL blocks) belongs to the last segment."""
        recs = list(reversed(self.tape))
        per_rec = [self._record_params(r) for r in recs]
        seen = set()
        for ps in per_rec:
            for q in ps:
                seen.add(id(q))
        rest = [q for q in self.param_list if id(q) not in seen]         # embedding path
        total = sum(q.numel() for ps in per_rec for q in ps) + sum(q.numel() for q in rest)
        want = max(1, min(self.N_SEGMENTS, len(recs)))
        segs, start, acc, cur = [], 0, 0, []
        for i, ps in enumerate(per_rec):
            cur += ps
            acc += sum(q.numel() for q in ps)
            last = i == len(recs) - 1
            if last or (len(segs) < want - 1 and acc >= total * (len(segs) + 1) / want):
                segs.append([start, rec_ends[i], cur + (rest if last else [])])
                start, cur = rec_ends[i], []
        self.bsegs = segs

    @staticmethod
    def _only_owner(t: torch.Tensor) -> bool:
        """True when nothing but ``t`` itself references its storage: no ``param.grad`` adopted by autograd, no tensor returned by
        ``torch.autograd.grad`` or kept by the caller across ``zero_grad(set_to_none=True)``, no DDP bucket view."""
        use_count = getattr(torch._C, "_storage_Use_Count", None)
        if use_count is None:
            # a private torch symbol: without it no buffer is ever recycled (correct, but every backward allocates and zero-fills a
            # fresh 0.95 GB gradient buffer) -- say so once instead of silently doubling the training step's memory traffic
            if not getattr(_Plan, "_warned_use_count", False):
                _Plan._warned_use_count = True
                import warnings
                warnings.warn("bbdm_amd: torch._C._storage_Use_Count is not available in this torch build; the flat gradient "
                              "buffer of the training plan is re-allocated on every backward pass instead of being recycled")
            return False                                   # cannot tell: never recycle
        return use_count(t.untyped_storage()._cdata) <= 2  # t + the temporary storage handle of this query

    def _pick_flat_grad(self):
        """The flat gradient buffer of this backward pass.  Up to two persistent buffers instead of a fresh 0.95 GB allocation
        per call -- but a buffer is recycled only when NOTHING else holds its storage: the gradients handed to autograd are views
        of it (``param.grad`` after the first backward, the tensors ``torch.autograd.grad`` returns, anything the caller keeps for
        logging or a gradient penalty), and zeroing it under them would silently rewrite those tensors."""
        if not hasattr(self, "_flat_bufs"):
            self._flat_bufs = []
        self._flat_grad = None                              # (drop this plan's own second handle before counting owners)
        for fb in self._flat_bufs:
            if self._only_owner(fb):
                return fb.zero_()
        fresh = torch.zeros(self.grad_total, dtype=torch.float32, device=self.device)
        if len(self._flat_bufs) < 2:
            self._flat_bufs.append(fresh)
        else:                                               # both still referenced: keep the newer pair candidate
            self._flat_bufs[0], self._flat_bufs[1] = self._flat_bufs[1], fresh
        return fresh

    def grad_view(self, q):
        off = self.grad_off[id(q)]
        return self._flat_grad[off:off + q.numel()].view_as(q)

    def backward_begin(self, dout: torch.Tensor):
        with _lib.device_guard(self.device):
            self._flat_grad = self._pick_flat_grad()
            self.dout_nchw.copy_(dout)
            stream = _lib.current_stream(self.device)
            if getattr(self, "_dconvs_forked", False):          # (re-packed beside the forward, see _run)
                torch.cuda.current_stream(self.device).wait_stream(self._side_stream)
                self._dconvs_forked = False
            for pk in self.dconvs:
                pk.refresh(stream)
            if self._h2_dy_slots:                               # the measured maxima of this backward's gradient tensors accumulate from 0
                self._h2_bounds.t[self._h2_bounds.t.numel() - self._h2_dy_slots:].zero_()

    def backward_segment(self, k: int, need_dx: bool = False):
        """Enqueue segment ``k`` of the gradient plan (0 = head side).  The last segment also runs the embedding path and,
        on request, the input gradient; returns the NHWC input-gradient view or None."""
        with _lib.device_guard(self.device):
            return self._backward_segment(k, need_dx)

    def _backward_segment(self, k, need_dx):
        lo, hi, _ = self.bsegs[k]
        last = k == len(self.bsegs) - 1
        stream = _lib.current_stream(self.device)
        lib = self.lib
        ops = self.bops[lo:hi] + (self.bops_x0 if (last and need_dx) else [])
        check = _lib.check
        prof = self.m.op_profile
        # second-stream ranges of this segment (forward: _launch_forward): op index -> stream, forks and joins
        on_side, forks, joins = {}, set(), set()
        if prof is None and self._bside_ranges and self.device.type == "cuda":
            for k0, k1, kj in self._bside_ranges:
                if lo <= k0 and kj < hi:
                    forks.add(k0 - lo)
                    joins.add(kj - lo)
                    for j in range(k0, k1):
                        on_side[j - lo] = True
            if forks:
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(self.device)
                side_s, main_s = self._side_stream, torch.cuda.current_stream(self.device)
        for i, (name, args) in enumerate(ops):
            fn = getattr(lib, getattr(name, "entry", name))
            if i in forks:
                side_s.wait_stream(main_s)
            if i in joins:
                main_s.wait_stream(side_s)
            if i in on_side:
                rc = fn(*(a.resolve() if hasattr(a, "resolve") else a for a in args), side_s.cuda_stream)
            elif prof is None:
                rc = fn(*(a.resolve() if hasattr(a, "resolve") else a for a in args), stream)
            else:                                   # per-op HIP events (bench.py's roofline leg), as in _launch_forward
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*(a.resolve() if hasattr(a, "resolve") else a for a in args), stream)
                e1.record()
                prof.append((_OpName(str(name) + ":bwd", getattr(name, "entry", str(name))), e0, e1, self._algorithmic_flops(str(name), args)))
            if rc != 0:
                check(rc, name)
        # weight gradients computed against a channel-padded input (stem, cross-attention to_k / to_v on a 3-channel context) live
        # in a padded scratch tensor: cut them into the flat gradient BEFORE this segment's views are handed to autograd (DDP's
        # reducer hooks and gradient accumulation read them as soon as the segment's node returns)
        for w, t, at in self._padded_wgrads:
            if lo <= at < hi:
                self.grad_view(w).copy_(t[:, : w.shape[1]].reshape(w.shape))      # (Linear weights are 2-D: ks = 1)
        if not last:
            return None
        return self._backward_embedding(stream, need_dx)

    def context_token_grad(self) -> Optional[torch.Tensor]:
        """Gradient reaching the context through the cross-attention keys / values, [N, Cc, Hc, Wc] (None without
        SpatialTransformer blocks); valid after the last backward segment.  The context's other gradient -- it is also
        concatenated to the UNet input (openaimodel.py:741-742) -- is part of the input gradient."""
        v = self.dctx_tokens
        if v is None:
            return None
        t = v.buf.tensor[v.off: v.off + v.N * v.H * v.W * v.ld].view(v.N, v.H, v.W, v.ld)
        return t[..., :self.ctx_in.shape[1]].permute(0, 3, 1, 2)

    def run_backward(self, dout: torch.Tensor, need_dx: bool):
        """Whole gradient plan in one go.  Returns (flat parameter gradient, d input NHWC view or None)."""
        self.backward_begin(dout)
        dx_in = None
        for k in range(len(self.bsegs)):
            dx_in = self.backward_segment(k, need_dx)
        return self._flat_grad, dx_in

    def _backward_embedding(self, stream, need_dx: bool):
        m, N = self.m, self.N
        flat = self._flat_grad
        gslice = self.grad_view
        # ---- embedding path: film projections -> time_embed.2 -> time_embed.0 ------------------------------------
        mc, ted = m.model_channels, 4 * m.model_channels
        call = _lib.call
        te0, te2 = m.time_embed[0], m.time_embed[2]
        first = True
        R = self.emb_rows
        # (the FiLM gradients land in the flat buffer itself: its first film_total x (4 mc + 1) floats, see _emit_backward)
        dfilm_w = flat[: self.film_total * ted].view(self.film_total, ted)
        dfilm_b = flat[self._film_b_off: self._film_b_off + self.film_total]
        for r0 in range(0, N, R):
            r = min(R, N - r0)
            tgt_w = dfilm_w if first else torch.empty_like(dfilm_w)
            tgt_b = dfilm_b if first else torch.empty_like(dfilm_b)
            call("bbdm_linear_bwd_f32", self.dfilm.data_ptr() + 4 * r0 * self.film_total,
                 self.emb.data_ptr() + 4 * r0 * ted, self.film_w.data_ptr(), self.d_emb.data_ptr() + 4 * r0 * ted,
                 tgt_w.data_ptr(), tgt_b.data_ptr(), self._lin_ws.data_ptr(), r, ted, self.film_total, 1, stream)
            g2w = gslice(te2.weight) if first else torch.empty_like(te2.weight)
            g2b = gslice(te2.bias) if first else torch.empty_like(te2.bias)
            call("bbdm_linear_bwd_f32", self.d_emb.data_ptr() + 4 * r0 * ted, self.e1.data_ptr() + 4 * r0 * ted,
                 te2.weight.data_ptr(), self.d_e1.data_ptr() + 4 * r0 * ted, g2w.data_ptr(), g2b.data_ptr(),
                 self._lin_ws.data_ptr(), r, ted, ted, 1, stream)
            g0w = gslice(te0.weight) if first else torch.empty_like(te0.weight)
            g0b = gslice(te0.bias) if first else torch.empty_like(te0.bias)
            call("bbdm_linear_bwd_f32", self.d_e1.data_ptr() + 4 * r0 * ted, self.e0.data_ptr() + 4 * r0 * mc,
                 te0.weight.data_ptr(), None, g0w.data_ptr(), g0b.data_ptr(), self._lin_ws.data_ptr(), r, mc, ted, 0,
                 stream)
            if not first:                      # batch > 64: sum the per-chunk weight gradients
                dfilm_w += tgt_w; dfilm_b += tgt_b
                gslice(te2.weight).add_(g2w); gslice(te2.bias).add_(g2b)
                gslice(te0.weight).add_(g0w); gslice(te0.bias).add_(g0b)
            first = False
        dx_in = None
        if need_dx:
            v = self.dx0
            dx_in = v.buf.tensor[: v.N * v.H * v.W * v.ld].view(v.N, v.H, v.W, v.ld)
        return dx_in

    # ---- execution ------------------------------------------------------------------------------------------------
    def _bind(self):
        lib = self.lib
        self._bound_names = [name for name, _ in self.ops]       # (_OpName objects: bench.py reads the entry point a launch is bound to)
        self._bound = [(getattr(lib, getattr(name, "entry", name)), tuple(a.resolve() if hasattr(a, "resolve") else a for a in args))
                       for name, args in self.ops]
        if self._h2_layers:       # csrc/groupnorm.hip: H2GnLayer {gamma, beta, film_off, C, zmax, gain}
            import struct
            raw = b"".join(struct.pack("QQiiff", g.data_ptr(), b.data_ptr(), fo, C, z, 1.0) for g, b, fo, C, z in self._h2_layers)
            self._h2_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)

    def _refresh_weights(self, stream):
        key = tuple(p.data_ptr() for p in self._params)
        if key != self._param_key:
            self._bind()
            self._param_key = key
        for pc in self.convs:
            pc.refresh(stream)
        fkey = tuple((rb.emb_layers[1].weight.data_ptr(), _ver(rb.emb_layers[1].weight),
                      rb.emb_layers[1].bias.data_ptr(), _ver(rb.emb_layers[1].bias)) for rb in self.resblocks)
        if fkey != self._film_key:
            off = 0
            for rb in self.resblocks:
                lin = rb.emb_layers[1]
                n = lin.out_features
                self.film_w[off:off + n].copy_(lin.weight.detach())
                self.film_b[off:off + n].copy_(lin.bias.detach())
                off += n
            if self.film_wp is not None:
                _lib.call("bbdm_linear_pack_f32", self.film_w.data_ptr(), self.film_wp.data_ptr(), self.film_total,
                          self.film_w.shape[1], stream)
            self._film_key = fkey

    def _launch_embedding(self, stream):
        """Statistics reset + the embedding path (timestep embedding, time_embed MLP, all FiLM projections) on ``stream`` -- which must
        be torch's current stream (the reset is a torch op)."""
        m, N = self.m, self.N
        self.stats.zero_()
        mc, ted = m.model_channels, 4 * m.model_channels
        call = _lib.call
        te0, te2 = m.time_embed[0], m.time_embed[2]
        call("bbdm_timestep_embedding_f32", self.t_buf.data_ptr(), m.freqs(self.device).data_ptr(),
             self.e0.data_ptr(), N, mc, stream)
        R = self.emb_rows
        for r0 in range(0, N, R):           # bbdm_linear_f32 stages its rows in LDS: <= emb_rows per call
            r = min(R, N - r0)
            # e1 holds the PRE-activation of time_embed.0; its SiLU is applied as the next layer's act_in
            call("bbdm_linear_f32", self.e0.data_ptr() + 4 * r0 * mc, te0.weight.data_ptr(), te0.bias.data_ptr(),
                 self.e1.data_ptr() + 4 * r0 * ted, r, mc, ted, 0, 0, stream)
            call("bbdm_linear_f32", self.e1.data_ptr() + 4 * r0 * ted, te2.weight.data_ptr(), te2.bias.data_ptr(),
                 self.emb.data_ptr() + 4 * r0 * ted, r, ted, ted, 1, 0, stream)
            if self.film_wp is None:
                call("bbdm_linear_f32", self.emb.data_ptr() + 4 * r0 * ted, self.film_w.data_ptr(),
                     self.film_b.data_ptr(), self.film.data_ptr() + 4 * r0 * self.film_total, r, ted, self.film_total,
                     1, 0, stream)
        if self.film_wp is not None:
            for r0 in range(0, N, 32):      # (<= 32 rows per call of the packed kernel)
                call("bbdm_linear_packed_f32", self.emb.data_ptr() + 4 * r0 * ted, self.film_wp.data_ptr(), self.film_b.data_ptr(),
                     self.film.data_ptr() + 4 * r0 * self.film_total, min(32, N - r0), ted, self.film_total, 1, 0, stream)
        if self._h2_layers:
            # the bounds of every GroupNorm-fed tile GEMM on the fp16-pair planes, from gamma / beta / this step's FiLM vector
            call("bbdm_h2_gn_bounds_f32", self._h2_table.data_ptr(), len(self._h2_layers), self.film.data_ptr() if self.film_total else None,
                 self.film_total, N, self._h2_bounds.t.data_ptr(), stream)

    def _launch_forward(self, stream, prof=None):
        """Enqueue one forward (statistics reset, embedding path, the op list) on ``stream``."""
        self._launch_embedding(stream)
        check = _lib.check
        if prof is None and self._side_ranges and self.device.type == "cuda":
            # fork / join around the side ranges (captured into the hipGraph as parallel branches; eager launches behave the same)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
            side, main = self._side_stream, torch.cuda.current_stream(self.device)
            forks = {k0: k1 for k0, k1, _ in self._side_ranges}
            joins = {kj for _, _, kj in self._side_ranges}
            k, n = 0, len(self._bound)
            while k < n:
                if k in forks:
                    side.wait_stream(main)
                    for j in range(k, forks[k]):
                        fn, args = self._bound[j]
                        rc = fn(*args, side.cuda_stream)
                        if rc != 0:
                            check(rc, fn.__name__)
                    k = forks[k]
                    continue
                if k in joins:
                    main.wait_stream(side)
                fn, args = self._bound[k]
                rc = fn(*args, stream)
                if rc != 0:
                    check(rc, fn.__name__)
                k += 1
        elif prof is None:
            for fn, args in self._bound:
                rc = fn(*args, stream)
                if rc != 0:
                    check(rc, fn.__name__)
        else:
            # per-op HIP events on the launch stream (bench.py's roofline leg); (name, start, stop, flops)
            for (fn, args), fl, opname in zip(self._bound, self.op_flops, self._bound_names):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args, stream)
                e1.record()
                if rc != 0:
                    check(rc, fn.__name__)
                prof.append((opname, e0, e1, fl))

    def _want_graph(self) -> bool:
        """Inference plans replay their ~200 launches as ONE hipGraph: at the small latents it removes most of the step (launch-bound),
        at 256^2 / batch 16 -- launches of milliseconds -- still 1.4 ms of the 114 (the ~1.5 us boundaries between dependent launches and
        the host's per-call work; measured round 3).  ``UNetModel.hip_graph`` = True / False overrides.  Training plans launch kernel by
        kernel (replaying their forward and backward segments as graphs measured 0.6 % on C4 in round 3 and was removed in round 5)."""
        pref = self.m.hip_graph
        if pref is not None:
            return bool(pref)
        if self.device.type != "cuda" or self.training:
            return False
        return True

    def holds_input(self, x: torch.Tensor):
        """``x_in`` now holds the value of ``x`` (the fused bridge kernel wrote x_next there as well: csrc/bridge.hip)."""
        v = _ver(x)
        self._x_src = None if isinstance(v, _NoVersion) else (x, v)

    def run(self, x, t, ctx, out=None, borrow=False):
        with _lib.device_guard(self.device):        # NULL-stream launches follow the current device (see _lib.device_guard)
            return self._run(x, t, ctx, out, borrow)

    def _run(self, x, t, ctx, out=None, borrow=False):
        m = self.m
        self.generation = getattr(self, "generation", 0) + 1
        stream = _lib.current_stream(self.device)
        self._refresh_weights(stream)
        if (self.training and self.m.side_stream_train and self.m.side_stream_dgrad_pack and self.device.type == "cuda"
                and getattr(self, "dconvs", None)):
            # the data-gradient orientation of the weights (G g' G^T planes, packed 1x1 / direct operands) is not read before the
            # backward: after an optimizer step it is re-packed on the second stream, beside this forward (backward_begin joins)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
            side = self._side_stream
            side.wait_stream(torch.cuda.current_stream(self.device))
            for pk in self.dconvs:
                pk.refresh(side.cuda_stream)
            self._dconvs_forked = True
        # Inputs: skip the copy of a tensor this plan already holds -- the x_next the previous sampling step also wrote into x_in
        # (holds_input), the conditioning image that does not change during a sampling loop.  Identity AND version are checked and the
        # source is kept alive, so neither an in-place edit nor a recycled allocation can be mistaken for it.
        src = getattr(self, "_x_src", None)
        if not (src is not None and src[0] is x and src[1] == _ver(x)):
            self.x_in.copy_(x)
        self._x_src = None
        if self.ctx_in is not None:
            csrc = getattr(self, "_ctx_src", None)
            if not (csrc is not None and csrc[0] is ctx and csrc[1] == _ver(ctx)):
                self.ctx_in.copy_(ctx)
                v = _ver(ctx)
                self._ctx_src = None if isinstance(v, _NoVersion) else (ctx, v)
        if isinstance(t, int):
            self.t_buf.fill_(t)                                   # (p_sample: one timestep for the whole batch)
        else:
            self.t_buf.copy_(t)
        prof = m.op_profile
        if prof is None and self._want_graph() and not getattr(self, "_graph_failed", False):
            # weights / parameter pointers are checked above on every call; a change re-captures
            gkey = (self._param_key,) + tuple(p.data_ptr() for p in m.time_embed.parameters())
            if self._graph is None or self._graph_key != gkey:
                self._launch_forward(stream)                     # eager warm-up (sets kernel attributes, fills caches)
                torch.cuda.current_stream(self.device).synchronize()
                try:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: other threads of the process may touch the runtime while this one captures (the RCCL watchdog of
                    # a torch.distributed job polls its events: under the default mode that aborts the capture)
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._launch_forward(_lib.current_stream(self.device))
                    self._graph, self._graph_key = g, gkey
                except RuntimeError as e:          # capture refused (e.g. another thread inside the runtime): launch by launch, same kernels
                    import warnings
                    warnings.warn(f"bbdm_amd: hipGraph capture of the forward failed ({e}); launching kernel by kernel")
                    self._graph, self._graph_failed = None, True
                    torch.cuda.synchronize(self.device)
                    self._launch_forward(stream)
            if self._graph is not None:
                self._graph.replay()
        else:
            self._launch_forward(stream, prof)
        if out is None:
            return self.out_nchw if borrow else self.out_nchw.clone()
        out.copy_(self.out_nchw)
        return out
