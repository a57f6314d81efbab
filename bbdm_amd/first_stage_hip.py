"""The frozen VQGAN first stage on the HIP kernels.

``bbdm_amd.first_stage.VQModel`` (plain PyTorch-ROCm, what BASELINE.json's north_star asks for at minimum) stays the
parameter container: same constructor, ``state_dict`` and call surface as the reference's ``model.VQGAN.vqgan.VQModel``.
:class:`VQModel` here derives from it and adds two whole-pipeline entry points that ``LatentBrownianBridgeModel`` uses when
they exist:

* ``encode_latent(x, quant_conv=True)``      = ``quant_conv(encoder(x))``            (LatentBrownianBridgeModel.py:73-85)
* ``decode_latent(z, quant_conv_first=False)`` = ``decode(quantize([quant_conv](z)))``  (LatentBrownianBridgeModel.py:87-100)

Each compiles, per input shape, a static list of C-ABI calls over pre-allocated NHWC buffers -- the execution-plan machinery
of ``bbdm_amd.unet`` (``_Plan``'s emitters: GroupNorm(eps 1e-6) -> SiLU folded into the consuming convolution or Winograd
input transform, Winograd / bf16x3 tile GEMMs for the wide 3x3 layers, the narrow-Cout kernel for the 3-channel head) plus
what only the VQGAN needs (csrc/firststage.hip):

* ``AttnBlock`` (model/VQGAN/model.py:140-192): ONE head over all 128..512 channels -> scores materialised by the batched
  activation GEMM (``bbdm_gemm_pack_b_f32`` / ``bbdm_gemm_batched_f32``), ``bbdm_softmax_rows_f32``, second GEMM;
* ``Downsample`` (model.py:56-75): stride-2 conv on a (0,1,0,1)-padded input = the stride-1 'same' conv at the odd
  positions (``bbdm_groupnorm_apply_f32`` resample mode 4);
* ``VectorQuantizer2`` (quantize.py:271-286): ``bbdm_vq_nearest_f32`` (indices bit-equal to the reference expression's).

Inference only (the first stage is frozen: LatentBrownianBridgeModel.py:23-27).  On CPU tensors, or under autograd, the
inherited PyTorch modules run -- that is the reference's own path, not a fallback of the HIP one: ``encode_latent`` /
``decode_latent`` themselves raise on non-GPU tensors like every other bbdm_amd entry point.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from . import first_stage as FS
from .unet import _Plan, _TensorRef, _View, _round4

__all__ = ["VQModel"]


class _Flags:
    """The execution-plan switches ``_Plan``'s emitters read from their owner (``UNetModel`` for the UNet)."""

    def __init__(self):
        self.winograd = 6
        self.winograd_fuse_groupnorm = True
        self.winograd_small = True
        self.upsample_phases = True
        self.gemm_bf3 = True
        self.gemm_bf3p = True
        self.gemm_h2 = True                 # tile GEMMs of the GroupNorm-fed 3x3 layers on the fp16-pair planes (csrc/h2_split.h), as in the UNet
        self.gemm_h2_train = 0
        self.conv1x1_h2 = False             # (its 1x1 layers -- the AttnBlock projections -- keep bf16x3)
        self.attn_h2 = False                # (... and its single-head attention the one-launch bf16x3 kernel)
        self.fuse_stats = True
        self.bf3_min_tiles = 256
        self.conv1x1_small = True
        self.gn_in_transform = 1024
        self.fp32_v_max_cout = 128
        self.upsample_f72 = True
        self.winograd_wgrad = 0             # (inference plans only)
        self.winograd_train8 = 0
        self.winograd8_min_tiles = 512
        self.side_stream_min_macs = 0
        self.side_stream_max_macs = 0
        self.side_stream_max_pixels = 0
        self.side_stream_train = False
        self.side_stream_wgrad = False
        self.side_stream_dgrad_pack = False
        self.side_stream_wgrad_min_macs = 0
        self.hip_graph = False
        self.op_profile = None


class _FSPlan(_Plan):
    """Static schedule of C-ABI calls for one direction of the first stage at one input shape."""

    def __init__(self, vq: "VQModel", kind: str, N: int, H: int, W: int, device, quant_conv: bool):
        self._init_state(vq._flags, N, device, training=False)
        self.vq, self.kind = vq, kind
        f32 = dict(dtype=torch.float32, device=device)
        if kind == "encode":
            enc = vq.encoder
            cin = enc.conv_in.in_channels
            self.x_in = torch.empty(N, cin, H, W, **f32)
            x0 = self._new(N, H, W, _round4(cin))
            self._op("bbdm_nchw_to_nhwc_f32", _TensorRef(self.x_in), cin, None, 0, x0, x0.ld, x0.C, N, H, W)
            h = self._new(N, H, W, enc.conv_in.out_channels)
            self._emit_conv(x0, enc.conv_in, None, h)
            for level in enc.down:
                h = self._level(level, h)
                if hasattr(level, "downsample"):
                    h = self._down(level.downsample, h)
            h = self._middle(enc.mid, h)
            zc = enc.conv_out.out_channels
            if quant_conv:
                z = self._head(h, enc.norm_out, enc.conv_out, nchw=False)
                oc = vq.quant_conv.out_channels
                zq = self._new(N, z.H, z.W, _round4(oc))
                self._emit_conv(z, vq.quant_conv, None, _View(zq.buf, 0, zq.ld, N, z.H, z.W, oc))
                self.out_nchw = torch.empty(N, oc, z.H, z.W, **f32)
                self._op("bbdm_nhwc_to_nchw_f32", zq, zq.ld, _TensorRef(self.out_nchw), N, z.H, z.W, oc)
            else:
                self.out_nchw = torch.empty(N, zc, h.H, h.W, **f32)
                self._head(h, enc.norm_out, enc.conv_out, nchw=True)
        else:
            dec = vq.decoder
            zc_in = vq.quant_conv.in_channels if quant_conv else vq.quantize.e_dim
            self.x_in = torch.empty(N, zc_in, H, W, **f32)
            z = self._new(N, H, W, _round4(zc_in))
            self._op("bbdm_nchw_to_nhwc_f32", _TensorRef(self.x_in), zc_in, None, 0, z, z.ld, z.C, N, H, W)
            e_dim, n_e = vq.quantize.e_dim, vq.quantize.n_e
            if quant_conv:                              # latent_before_quant_conv: the bridge ran on the pre-quant_conv latent
                zz = self._new(N, H, W, _round4(e_dim))
                self._emit_conv(z, vq.quant_conv, None, _View(zz.buf, 0, zz.ld, N, H, W, e_dim))
                z = zz
            # nearest codebook entry per latent pixel (zero channel padding of the destination stays zero: it is never written)
            zq = self._new(N, H, W, _round4(e_dim))
            self.indices = torch.empty(N * H * W, dtype=torch.int64, device=device)
            self._op("bbdm_vq_nearest_f32", z, z.ld, self._pref(vq.quantize.embedding.weight), _TensorRef(self.indices), zq,
                     zq.ld, N * H * W, n_e, e_dim)
            self._zero_pad = (zq, e_dim)
            pq = vq.post_quant_conv
            hq = self._new(N, H, W, _round4(pq.out_channels))
            self._emit_conv(zq, pq, None, _View(hq.buf, 0, hq.ld, N, H, W, pq.out_channels))
            self._zero_pad2 = (hq, pq.out_channels)
            h = self._new(N, H, W, dec.conv_in.out_channels)
            self._emit_conv(hq, dec.conv_in, None, h)
            h = self._middle(dec.mid, h)
            for level in reversed(dec.up):
                h = self._level(level, h)
                if hasattr(level, "upsample"):
                    h = self._up(level.upsample, h)
            if dec.give_pre_end:
                self.out_nchw = torch.empty(N, h.C, h.H, h.W, **f32)
                self._op("bbdm_nhwc_to_nchw_f32", h, h.ld, _TensorRef(self.out_nchw), N, h.H, h.W, h.C)
            else:
                self.out_nchw = torch.empty(N, dec.conv_out.out_channels, h.H, h.W, **f32)
                self._head(h, dec.norm_out, dec.conv_out, nchw=True)
        self._allocate()
        # channel-padded buffers that are only partially written by their producer: the padding must read as zero
        for b in self.bufs:
            b.tensor.zero_()

    # ---- layers ----------------------------------------------------------------------------------------------------
    def _res(self, rb: FS.ResnetBlock, x: _View) -> _View:
        """ResnetBlock.forward (model/VQGAN/model.py:117-137; temb is None in the VQGAN)."""
        a, pre1 = self._gn_input(x, rb.norm1, None, silu=1, name="A", consumer=rb.conv1)
        h1 = self._tmp("H1", self.N, x.H, x.W, rb.conv1.out_channels)
        self._emit_conv(a, rb.conv1, None, h1, pre=pre1)
        a2, pre2 = self._gn_input(h1, rb.norm2, None, silu=1, name="A2", consumer=rb.conv2)
        out = self._new(self.N, x.H, x.W, rb.conv2.out_channels)
        if hasattr(rb, "nin_shortcut"):
            self._emit_conv(x, rb.nin_shortcut, None, out)
            self._emit_conv(a2, rb.conv2, out, out, pre=pre2)
        else:
            self._emit_conv(a2, rb.conv2, x, out, pre=pre2)
        return out

    def _attn(self, ab: FS.AttnBlock, x: _View) -> _View:
        """AttnBlock.forward (model.py:160-192): softmax(q k^T c^-1/2) v with ONE head over all c channels."""
        N, H, W, C = self.N, x.H, x.W, x.C
        T = H * W
        a, pre = self._gn_input(x, ab.norm, None, silu=0, name="A")
        q, k, v = (self._tmp(nm, N, H, W, C) for nm in ("FS_Q", "FS_K", "FS_V"))
        for mod, dst in ((ab.q, q), (ab.k, k), (ab.v, v)):
            self._emit_conv(a, mod, None, dst, pre=pre)
        lib = self.lib
        pk = self._tmp("FS_PK", 1, 1, 1, N * max(lib.bbdm_gemm_packed_b_floats(T, C), lib.bbdm_gemm_packed_b_floats(C, T)))
        s = self._tmp("FS_S", N, 1, T, T)
        o = self._tmp("FS_O", N, H, W, C)
        self._op("bbdm_gemm_pack_b_f32", k, k.ld, T * k.ld, pk, N, T, C, 0)                       # B = k [T x C]: out = q k^T
        self._op("bbdm_gemm_batched_f32", q, q.ld, T * q.ld, pk, s, T, T * T, N, T, C, T)
        self._op("bbdm_softmax_rows_f32", s, T, N * T, T, float(int(C) ** (-0.5)))
        self._op("bbdm_gemm_pack_b_f32", v, v.ld, T * v.ld, pk, N, C, T, 1)                       # B = v [T x C]: out = w v
        self._op("bbdm_gemm_batched_f32", s, T, T * T, pk, o, o.ld, T * o.ld, N, T, T, C)
        out = self._new(N, H, W, C)
        self._emit_conv(o, ab.proj_out, x, out)
        return out

    def _level(self, level, h: _View) -> _View:
        for i, blk in enumerate(level.block):
            h = self._res(blk, h)
            if len(level.attn):
                h = self._attn(level.attn[i], h)
        return h

    def _middle(self, mid, h: _View) -> _View:
        return self._res(mid.block_2, self._attn(mid.attn_1, self._res(mid.block_1, h)))

    def _down(self, ds: FS.Downsample, x: _View) -> _View:
        """Downsample.forward (model.py:68-75): pad (0,1,0,1) + stride-2 3x3 conv == the odd positions of the pad-1 stride-1
        conv; or a 2x2 average pool."""
        N = self.N
        if x.H % 2 or x.W % 2:
            raise NotImplementedError("bbdm_amd.first_stage_hip: Downsample needs even H, W")
        out = self._new(N, x.H // 2, x.W // 2, x.C)
        if ds.with_conv:
            full = self._tmp("DSF", N, x.H, x.W, x.C)
            self._emit_conv(x, ds.conv, None, full)
            self._op("bbdm_groupnorm_apply_f32", full, full.ld, None, None, None, None, 0, out, out.ld, N, x.H, x.W, x.C, 1, 0.0,
                     0, 4)
        else:
            self._op("bbdm_groupnorm_apply_f32", x, x.ld, None, None, None, None, 0, out, out.ld, N, x.H, x.W, x.C, 1, 0.0, 0, 1)
        return out

    def _up(self, us: FS.Upsample, x: _View) -> _View:
        """Upsample.forward (model.py:47-53): nearest x2 (folded into the Winograd input transform when the conv takes that
        path) then the optional 3x3 conv."""
        N = self.N
        out = self._new(N, 2 * x.H, 2 * x.W, x.C)
        if us.with_conv and self._winograd_ok(us.conv, 2 * x.H, 2 * x.W, x.C):
            self._emit_conv(x, us.conv, None, out, upsample=True)
        elif us.with_conv:
            u = self._gn_apply(x, None, None, 0, 2, name="XR")
            self._emit_conv(u, us.conv, None, out)
        else:
            self._op("bbdm_groupnorm_apply_f32", x, x.ld, None, None, None, None, 0, out, out.ld, N, x.H, x.W, x.C, 1, 0.0, 0, 2)
        return out

    def _head(self, h: _View, norm, conv, nchw: bool) -> Optional[_View]:
        """norm_out -> swish -> conv_out (model.py:428-431,530-533).  Few output channels: the one-thread-per-pixel kernel with
        the normalisation folded into its staging (csrc/conv_igemm.hip: conv3x3_narrow_kernel)."""
        N = self.N
        cout = conv.out_channels
        narrow = cout <= 8 and N * h.H * h.W >= 4096
        a, pre = self._gn_input(h, norm, None, silu=1, name="A", consumer=None if narrow else conv, fuse_direct=narrow)
        if nchw:
            pc = self._conv(conv, a.C)
            self._conv_ws_need = max(self._conv_ws_need,
                                     self.lib.bbdm_conv_splitk_workspace_floats(N, a.H, a.W, a.C, cout, 3))
            self._op("bbdm_conv2d_nhwc_f32", a, a.ld, _TensorRef(pc.packed), self._pref(pc.bias), None, 0,
                     _TensorRef(self.out_nchw), 0, 1, self._conv_ws, self._conv_ws_floats, *pre, N, a.H, a.W, a.C, cout, 3)
            return None
        z = self._new(N, h.H, h.W, _round4(cout))
        self._emit_conv(a, conv, None, _View(z.buf, 0, z.ld, N, h.H, h.W, cout), pre=pre)
        return z

    # ---- execution ---------------------------------------------------------------------------------------------------
    def run(self, x: torch.Tensor) -> torch.Tensor:
        with _lib.device_guard(self.device):
            stream = _lib.current_stream(self.device)
            self._refresh_weights(stream)
            self.x_in.copy_(x)
            self.stats.zero_()
            if self._h2_layers:         # the GroupNorm bounds of the fp16-pair tile GEMMs (no FiLM in the first stage: gamma / beta alone)
                _lib.call("bbdm_h2_gn_bounds_f32", self._h2_table.data_ptr(), len(self._h2_layers), None, 0, self.N,
                          self._h2_bounds.t.data_ptr(), stream)
            check = _lib.check
            for fn, args in self._bound:
                rc = fn(*args, stream)
                if rc != 0:
                    check(rc, fn.__name__)
            return self.out_nchw.clone()


class VQModel(FS.VQModel):
    """``bbdm_amd.first_stage.VQModel`` + the two HIP pipelines (module docstring)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._flags = _Flags()
        self._fs_plans: Dict[tuple, _FSPlan] = {}

    def _apply(self, fn, *a, **k):
        self._fs_plans = {}                       # plans hold device pointers
        return super()._apply(fn, *a, **k)

    def _plan(self, kind, x, quant_conv) -> _FSPlan:
        N, _, H, W = x.shape
        key = (kind, N, H, W, x.device.index, bool(quant_conv), self._flags.winograd, self._flags.gemm_bf3, self._flags.gemm_bf3p)
        p = self._fs_plans.get(key)
        if p is None:
            if len(self._fs_plans) >= 4:           # a training loop alternates 2 encode shapes + 1 decode shape at most
                self._fs_plans.pop(next(iter(self._fs_plans)))
            p = self._fs_plans[key] = _FSPlan(self, kind, N, H, W, x.device, bool(quant_conv))
        return p

    @staticmethod
    def _prep(x):
        _lib.require_gpu(x)
        if x.dim() != 4:
            raise ValueError(f"expected [N, C, H, W], got {tuple(x.shape)}")
        return x.detach().float().contiguous()

    @torch.no_grad()
    def encode_latent(self, x: torch.Tensor, quant_conv: bool = True) -> torch.Tensor:
        """``quant_conv(encoder(x))`` (or ``encoder(x)`` alone) on the HIP kernels: [N, 3, H, W] -> [N, z, H/f, W/f]."""
        x = self._prep(x)
        if x.shape[1] != self.encoder.conv_in.in_channels:
            raise RuntimeError(f"expected {self.encoder.conv_in.in_channels} input channels, got {x.shape[1]}")
        return self._plan("encode", x, quant_conv).run(x)

    @torch.no_grad()
    def decode_latent(self, z: torch.Tensor, quant_conv_first: bool = False, return_indices: bool = False):
        """``decode(quantize(z))`` (with ``quant_conv`` first when the bridge ran before it) on the HIP kernels."""
        z = self._prep(z)
        want = self.quant_conv.in_channels if quant_conv_first else self.quantize.e_dim
        if z.shape[1] != want:
            raise RuntimeError(f"expected a {want}-channel latent, got {z.shape[1]}")
        plan = self._plan("decode", z, quant_conv_first)
        out = plan.run(z)
        if return_indices:
            return out, plan.indices.view(z.shape[0], z.shape[2], z.shape[3]).clone()
        return out
