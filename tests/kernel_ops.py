"""Tensor-level wrappers over the C-ABI (one function per entry point of include/bbdm_hip.h) -- TEST SCAFFOLDING.

They allocate outputs with torch, pass raw pointers + the current HIP stream, and raise on error.  Used by the kernel
unit tests and the tools/ benchmarks only; the product path (bbdm_amd/unet.py, model.py) calls the library directly with
pre-resolved pointers.  NHWC tensors here are plain contiguous ``[N, H, W, C]`` torch tensors (pitch = C) unless a wider
buffer + channel slice is passed explicitly via ``ld`` arguments.

Two back ends: the real library (GPU tensors; the default), or -- after ``use_emulator()`` -- the host build of the same
kernel sources running on tools/hipemu's CPU emulation of the HIP device model (CPU tensors; tests/test_emu_*.py).
"""
from __future__ import annotations

import os
import sys
from typing import Optional

import torch

from bbdm_amd import _lib as _real

_EMU = None          # ctypes handle of tools/hipemu/_build/libbbdm_emu.so once use_emulator() has been called


class _Lib:
    """The subset of bbdm_amd._lib the wrappers use, routed to the selected back end."""
    BBDMHipError = _real.BBDMHipError

    @staticmethod
    def load():
        return _EMU if _EMU is not None else _real.load()

    @staticmethod
    def call(name, *args):
        lib = _Lib.load()
        rc = getattr(lib, name)(*args)
        if rc != 0:
            msg = lib.bbdm_last_error()
            raise _real.BBDMHipError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")


_lib = _Lib


def use_emulator(on: bool = True):
    """Route every wrapper to the CPU-emulated host build (built on demand by tools/hipemu/build.py)."""
    global _EMU
    if not on:
        _EMU = None
        return None
    if _EMU is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "tools", "hipemu"))
        import build as hipemu_build
        _EMU = _real.bind(hipemu_build.build(asan=os.environ.get("HIPEMU_ASAN") == "1"))
    return _EMU


def _st(t):
    if not t.is_cuda:
        return None          # the emulator ignores the stream
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if t.is_cuda == (_EMU is not None):
            raise _real.BBDMHipError(f"kernel_ops: back end / tensor device mismatch ({t.device}, emulator={_EMU is not None})")
        if t.dtype not in (torch.float32, torch.float64, torch.int64):
            raise TypeError(f"unsupported dtype {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")


def nchw_to_nhwc(a: torch.Tensor, b: Optional[torch.Tensor] = None, cpad: Optional[int] = None) -> torch.Tensor:
    _chk(a, b)
    N, Ca, H, W = a.shape
    Cb = 0 if b is None else b.shape[1]
    cpad = cpad or (Ca + Cb + 3) // 4 * 4
    out = torch.empty(N, H, W, cpad, dtype=torch.float32, device=a.device)
    _lib.call("bbdm_nchw_to_nhwc_f32", a.data_ptr(), Ca, None if b is None else b.data_ptr(), Cb, out.data_ptr(),
              cpad, cpad, N, H, W, _st(a))
    return out


def nhwc_to_nchw(x: torch.Tensor, C: Optional[int] = None) -> torch.Tensor:
    _chk(x)
    N, H, W, ld = x.shape
    C = C or ld
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_nhwc_to_nchw_f32", x.data_ptr(), ld, out.data_ptr(), N, H, W, C, _st(x))
    return out


def pack_conv_weight(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    """OIHW (or [O, I, 1] conv1d) fp32 -> packed buffer for :func:`conv2d_nhwc`."""
    _chk(w)
    cout, cin = w.shape[0], w.shape[1]
    ks = w.shape[2] if w.dim() == 4 else 1
    cin_pad = cin_pad or (cin + 3) // 4 * 4
    n = _lib.load().bbdm_conv_packed_floats(cout, cin_pad, ks)
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), packed.data_ptr(), cout, cin, cin_pad, ks, _st(w))
    return packed


def conv2d_nhwc(x: torch.Tensor, packed_w: torch.Tensor, bias: Optional[torch.Tensor], cout: int, ks: int,
                residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                out_nchw: bool = False, split_k: bool = True, pre_scale: Optional[torch.Tensor] = None,
                pre_bias: Optional[torch.Tensor] = None, pre_silu: bool = False) -> torch.Tensor:
    """x: [N, H, W, CinPad] -> [N, H, W, cout] (or NCHW).  ``residual`` [N, H, W, cout] is added in the epilogue."""
    _chk(x, packed_w, bias, residual, out)
    N, H, W, cin_pad = x.shape
    if out is None:
        shape = (N, cout, H, W) if out_nchw else (N, H, W, cout)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    nws = _lib.load().bbdm_conv_splitk_workspace_floats(N, H, W, cin_pad, cout, ks) if split_k else 0
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    _lib.call("bbdm_conv2d_nhwc_f32", x.data_ptr(), cin_pad, packed_w.data_ptr(),
              None if bias is None else bias.data_ptr(), None if residual is None else residual.data_ptr(),
              0 if residual is None else residual.shape[-1], out.data_ptr(), 0 if out_nchw else out.shape[-1],
              1 if out_nchw else 0, None if ws is None else ws.data_ptr(), nws,
              None if pre_scale is None else pre_scale.data_ptr(), None if pre_bias is None else pre_bias.data_ptr(),
              0 if pre_scale is None else pre_scale.shape[-1], 1 if pre_silu else 0, N, H, W, cin_pad, cout, ks, _st(x))
    return out


def pack_winograd_weight(w: torch.Tensor, in_pad: Optional[int] = None, dgrad: bool = False, m: int = 2) -> torch.Tensor:
    """OIHW 3x3 fp32 -> G g G^T of F(m x m, 3x3) in the packed layout of :func:`conv3x3_winograd` (``dgrad``: the
    Cout -> Cin conv)."""
    _chk(w)
    cout, cin = w.shape[0], w.shape[1]
    src = cout if dgrad else cin
    in_pad = in_pad or (src + 3) // 4 * 4
    n = _lib.load().bbdm_winograd_packed_floats(m, cin if dgrad else cout, in_pad)
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.call("bbdm_winograd_pack_weight_f32", m, w.data_ptr(), packed.data_ptr(), cout, cin, in_pad, 1 if dgrad else 0,
              _st(w))
    return packed


def conv3x3_winograd(x: torch.Tensor, packed_w: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                     residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                     res_per_image: bool = False, m: int = 2, res_upsample: bool = False) -> torch.Tensor:
    """3x3 / stride 1 / pad 1 convolution through Winograd F(m x m, 3x3): x [N, H, W, CinPad] -> [N, H, W, cout].
    ``res_upsample``: the residual is [N, H/2, W/2, cout] and is added nearest-upsampled x2 (BBDM_CONV_RES_UPSAMPLE)."""
    _chk(x, packed_w, bias, residual, out)
    N, H, W, cin_pad = x.shape
    if out is None:
        out = torch.empty(N, H, W, cout, dtype=torch.float32, device=x.device)
    nws = _lib.load().bbdm_winograd_workspace_floats(m, N, H, W, cin_pad, cout)
    ws = torch.empty(max(1, nws), dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv3x3_winograd_f32", m, x.data_ptr(), cin_pad, packed_w.data_ptr(),
              None if bias is None else bias.data_ptr(), None if residual is None else residual.data_ptr(),
              0 if residual is None else residual.shape[-1], out.data_ptr(), out.shape[-1],
              2 if res_per_image else (4 if res_upsample else 0), ws.data_ptr(), N, H, W, cin_pad, cout, _st(x))
    return out


def groupnorm_coeffs(stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, hw: int,
                     film: Optional[torch.Tensor] = None, eps: float = 1e-5, groups: int = 32):
    """(scale, bias) [N, C] such that GN(x)[*(1+fs)+fb] == x * scale + bias per image and channel."""
    _chk(stats, gamma, beta, film)
    N, C = stats.shape[0], gamma.shape[0]
    sc = torch.empty(N, C, dtype=torch.float32, device=gamma.device)
    bi = torch.empty(N, C, dtype=torch.float32, device=gamma.device)
    _lib.call("bbdm_groupnorm_coeffs_f32", stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              None if film is None else film.data_ptr(), 0 if film is None else film.shape[1], sc.data_ptr(), bi.data_ptr(),
              C, N, hw, C, groups, float(eps), _st(gamma))
    return sc, bi


def new_stats(N: int, groups: int = 32, device=None) -> torch.Tensor:
    """A zeroed GroupNorm statistics accumulator (csrc/stats_acc.h: [N][G][2][4] 64-bit words, exact integer limbs)."""
    nbytes = _lib.load().bbdm_groupnorm_stats_bytes(N, groups)
    assert nbytes == N * groups * 2 * 4 * 8
    return torch.zeros(N, groups, 2, 4, dtype=torch.int64, device=device)


def read_stats(acc: torch.Tensor) -> torch.Tensor:
    """The accumulated (sum, sum of squares) of an accumulator as fp64 [N, G, 2] (bbdm_groupnorm_stats_read_f64)."""
    N, G = acc.shape[0], acc.shape[1]
    out = torch.empty(N, G, 2, dtype=torch.float64, device=acc.device)
    _lib.call("bbdm_groupnorm_stats_read_f64", acc.data_ptr(), out.data_ptr(), N, G, _st(acc))
    return out


def groupnorm_stats(x: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """-> the statistics ACCUMULATOR of x (read_stats() gives the fp64 sums)."""
    _chk(x)
    N, H, W, C = x.shape
    stats = new_stats(N, groups, x.device)
    _lib.call("bbdm_groupnorm_stats_f32", x.data_ptr(), C, stats.data_ptr(), N, H * W, C, groups, _st(x))
    return stats


def groupnorm_apply(x: torch.Tensor, stats: Optional[torch.Tensor], gamma: Optional[torch.Tensor],
                    beta: Optional[torch.Tensor], film: Optional[torch.Tensor] = None, eps: float = 1e-5,
                    silu: bool = False, resample: int = 0, groups: int = 32) -> torch.Tensor:
    """film: [N, 2C] (scale | shift).  resample: 0 none, 1 avg-pool 2x2, 2 nearest x2."""
    _chk(x, stats, gamma, beta, film)
    N, H, W, C = x.shape
    Ho, Wo = (H // 2, W // 2) if resample == 1 else ((2 * H, 2 * W) if resample == 2 else (H, W))
    y = torch.empty(N, Ho, Wo, C, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_groupnorm_apply_f32", x.data_ptr(), C, None if stats is None else stats.data_ptr(),
              None if gamma is None else gamma.data_ptr(), None if beta is None else beta.data_ptr(),
              None if film is None else film.data_ptr(), 0 if film is None else film.shape[1], y.data_ptr(), C,
              N, H, W, C, groups, float(eps), 1 if silu else 0, resample, _st(x))
    return y


def attention(qkv: torch.Tensor, heads: int, new_order: bool = False, return_lse: bool = False):
    """qkv: [N, T, 3*heads*ch] -> [N, T, heads*ch] (and the per-query log-sum-exp [N, heads, T] if asked)."""
    _chk(qkv)
    N, T, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty(N, T, C, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(N, heads, T, dtype=torch.float32, device=qkv.device) if return_lse else None
    _lib.call("bbdm_attention_f32", qkv.data_ptr(), C3, out.data_ptr(), C, None if lse is None else lse.data_ptr(),
              N, T, heads, C // heads, 1 if new_order else 0, _st(qkv))
    return (out, lse) if return_lse else out


def attention_planes(qkv: torch.Tensor, heads: int, new_order: bool = False, return_lse: bool = False, bound=None):
    """:func:`attention` in its two-launch form: bbdm_attention_kv_planes_f32 (K / V operand planes once per head) +
    bbdm_attention_planes_f32.  None where the library has no such form for the shape (bbdm_attention_kv_planes_bytes == 0).
    ``bound`` (a float >= max |qkv|, or a one-element device tensor): the fp16-pair form, bbdm_attention_kv_planes_h2_f32 +
    bbdm_attention_planes_h2_f32."""
    _chk(qkv)
    N, T, C3 = qkv.shape
    C = C3 // 3
    lib = _lib.load()
    h2 = bound is not None
    nbytes = (lib.bbdm_attention_kv_planes_h2_bytes if h2 else lib.bbdm_attention_kv_planes_bytes)(N, T, heads, C // heads)
    if nbytes == 0:
        return None
    if h2:
        bt = bound if torch.is_tensor(bound) else torch.tensor([float(bound)], dtype=torch.float32, device=qkv.device)
        planes = torch.empty(nbytes + 64, dtype=torch.uint8, device=qkv.device)
        planes[nbytes:] = 0xA5
        out = torch.empty(N, T, C, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(N, heads, T, dtype=torch.float32, device=qkv.device) if return_lse else None
        _lib.call("bbdm_attention_kv_planes_h2_f32", qkv.data_ptr(), C3, planes.data_ptr(), nbytes, N, T, heads, C // heads,
                  1 if new_order else 0, bt.data_ptr(), _st(qkv))
        _lib.call("bbdm_attention_planes_h2_f32", qkv.data_ptr(), C3, out.data_ptr(), C, None if lse is None else lse.data_ptr(),
                  N, T, heads, C // heads, 1 if new_order else 0, planes.data_ptr(), bt.data_ptr(), _st(qkv))
        assert bool((planes[nbytes:] == 0xA5).all())
        return (out, lse) if return_lse else out
    planes = torch.empty(nbytes + 64, dtype=torch.uint8, device=qkv.device)
    planes[nbytes:] = 0xA5                                   # (the launch may not write behind the size it reported)
    out = torch.empty(N, T, C, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(N, heads, T, dtype=torch.float32, device=qkv.device) if return_lse else None
    _lib.call("bbdm_attention_kv_planes_f32", qkv.data_ptr(), C3, planes.data_ptr(), nbytes, N, T, heads, C // heads,
              1 if new_order else 0, _st(qkv))
    _lib.call("bbdm_attention_planes_f32", qkv.data_ptr(), C3, out.data_ptr(), C, None if lse is None else lse.data_ptr(),
              N, T, heads, C // heads, 1 if new_order else 0, planes.data_ptr(), _st(qkv))
    assert bool((planes[nbytes:] == 0xA5).all())
    return (out, lse) if return_lse else out


def attention_bwd(qkv, out, dout, lse, heads: int, new_order: bool = False) -> torch.Tensor:
    _chk(qkv, out, dout, lse)
    N, T, C3 = qkv.shape
    C = C3 // 3
    dqkv = torch.empty_like(qkv)
    work = torch.empty(N * heads * T, dtype=torch.float32, device=qkv.device)
    _lib.call("bbdm_attention_bwd_f32", qkv.data_ptr(), C3, out.data_ptr(), C, dout.data_ptr(), C, lse.data_ptr(),
              work.data_ptr(), dqkv.data_ptr(), C3, N, T, heads, C // heads, 1 if new_order else 0, _st(qkv))
    return dqkv


def timestep_embedding(t: torch.Tensor, freqs: torch.Tensor, dim: int) -> torch.Tensor:
    _chk(t, freqs)
    emb = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    _lib.call("bbdm_timestep_embedding_f32", t.data_ptr(), freqs.data_ptr(), emb.data_ptr(), t.shape[0], dim, _st(t))
    return emb


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act_in: bool = False,
           act_out: bool = False) -> torch.Tensor:
    _chk(x, w, b)
    N, In = x.shape
    Out = w.shape[0]
    y = torch.empty(N, Out, dtype=torch.float32, device=x.device)
    for r0 in range(0, N, 64):
        r = min(64, N - r0)
        _lib.call("bbdm_linear_f32", x.data_ptr() + 4 * r0 * In, w.data_ptr(), None if b is None else b.data_ptr(),
                  y.data_ptr() + 4 * r0 * Out, r, In, Out, 1 if act_in else 0, 1 if act_out else 0, _st(x))
    return y


def linear_packed(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act_in: bool = False,
                  act_out: bool = False) -> torch.Tensor:
    """:func:`linear` on weights packed once (bbdm_linear_pack_f32 + bbdm_linear_packed_f32; N <= 32 rows, In % 32 == 0)."""
    _chk(x, w, b)
    N, In = x.shape
    Out = w.shape[0]
    lib = _lib.load()
    assert lib.bbdm_linear_packed_supported(N, In, Out)
    wp = torch.empty(lib.bbdm_linear_packed_bytes(Out, In), dtype=torch.uint8, device=x.device)
    _lib.call("bbdm_linear_pack_f32", w.data_ptr(), wp.data_ptr(), Out, In, _st(x))
    y = torch.empty(N, Out, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_linear_packed_f32", x.data_ptr(), wp.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), N, In, Out,
              1 if act_in else 0, 1 if act_out else 0, _st(x))
    return y


# ---------------------------------------------------------------------------------------------------------------
# backward (training path)
# ---------------------------------------------------------------------------------------------------------------
def pack_conv_weight_dgrad(w: torch.Tensor, cout_in: Optional[int] = None) -> torch.Tensor:
    """Packed transposed + flipped weights: conv2d_nhwc(dY, packed, None, cout=Cin, ks) is the data gradient."""
    _chk(w)
    cout, cin = w.shape[0], w.shape[1]
    ks = w.shape[2] if w.dim() == 4 else 1
    cout_in = cout_in or (cout + 3) // 4 * 4
    n = _lib.load().bbdm_conv_packed_dgrad_floats(cout, cin, cout_in, ks)
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.call("bbdm_conv_pack_weight_dgrad_f32", w.data_ptr(), packed.data_ptr(), cout, cin, cout_in, ks, _st(w))
    return packed


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, cin: int, cout: int, ks: int, with_bias: bool = False, ws_floats=None):
    """x: [N,H,W,CinPad], dy: [N,H,W,>=cout] -> dW [cout, cin, ks, ks] (cin = true, unpadded input channels);
    with_bias also returns db [cout] computed in the same pass."""
    _chk(x, dy)
    N, H, W, cin_pad = x.shape
    lib = _lib.load()
    need = lib.bbdm_conv_wgrad_workspace_floats(N, H, W, cin_pad, cout, ks)
    ws = torch.empty(need if ws_floats is None else max(1, ws_floats), dtype=torch.float32, device=x.device)   # (ws_floats: tests of the size check)
    dw = torch.empty(cout, cin_pad, ks, ks, dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if with_bias else None
    _lib.call("bbdm_conv_wgrad_f32", x.data_ptr(), cin_pad, dy.data_ptr(), dy.shape[-1], dw.data_ptr(),
              None if db is None else db.data_ptr(), ws.data_ptr(), ws.numel(), N, H, W, cin_pad, cout, ks, _st(x))
    dw = dw if cin == cin_pad else dw[:, :cin].contiguous()
    return (dw, db) if with_bias else dw


def conv3x3_winograd_wgrad(x: torch.Tensor, dy: torch.Tensor, cout: int, m: int, with_bias: bool = False):
    """x: [N,H,W,Cin], dy: [N,H,W,>=cout] -> dW [cout, Cin, 3, 3] (+ db) through the Winograd-domain weight gradient."""
    _chk(x, dy)
    N, H, W, cin = x.shape
    lib = _lib.load()
    ws = torch.empty(lib.bbdm_winograd_wgrad_workspace_floats(m, N, H, W, cin, cout), dtype=torch.float32, device=x.device)
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if with_bias else None
    _lib.call("bbdm_conv3x3_winograd_wgrad_f32", m, x.data_ptr(), cin, dy.data_ptr(), dy.shape[-1], dw.data_ptr(),
              None if db is None else db.data_ptr(), ws.data_ptr(), N, H, W, cin, cout, _st(x))
    return (dw, db) if with_bias else dw


def conv3x3_winograd_wgrad_bf3p(x: torch.Tensor, dy: torch.Tensor, cout: int, m: int, with_bias: bool = False):
    """The Winograd-domain weight gradient with both GEMM operands as transposed bf16 planes (csrc/gemm_bf3p.hip TN entry): x [N,H,W,Cin]
    (Cin % 32 == 0), dy [N,H,W,ld >= cout] -> (dW OIHW, db or None).  Stages: bbdm_winograd_input_bf3p_tr_f32 (as the training forward
    runs it), bbdm_winograd_dy_transform_bf3p_f32, bbdm_gemm_bf3p_tn_f32, bbdm_winograd_wgrad_finish_bias_f32 (dW + the column sums of dm11)."""
    _chk(x, dy)
    N, H, W, cin = x.shape
    lib = _lib.load()
    P, Tp = (m + 2) ** 2, lib.bbdm_winograd_tiles(m, N, H, W)
    T = N * -(-H // m) * -(-W // m)
    dev = x.device
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    Vp, Vt = u8(lib.bbdm_gemm_bf3p_a_bytes(P, Tp, cin)), u8(lib.bbdm_gemm_bf3p_tn_at_bytes(P, Tp, cin))
    dMt = u8(lib.bbdm_gemm_bf3p_tn_bt_bytes(P, Tp, cout))
    dm11 = torch.zeros(Tp, cout, dtype=torch.float32, device=dev)
    _lib.call("bbdm_winograd_input_bf3p_tr_f32", m, x.data_ptr(), cin, Vp.data_ptr(), None, None, 0, 0, 0, N, H, W, cin, Vt.data_ptr(), _st(x))
    _lib.call("bbdm_winograd_dy_transform_bf3p_f32", m, dy.data_ptr(), dy.shape[-1], dMt.data_ptr(), dm11.data_ptr(), N, H, W, cout, _st(x))
    assert lib.bbdm_gemm_bf3p_tn_supported(Tp, cin, cout)
    splits = lib.bbdm_gemm_bf3p_tn_splits(P, Tp, cin, cout)
    dU = torch.empty(splits * P * cin * cout, dtype=torch.float32, device=dev)
    _lib.call("bbdm_gemm_bf3p_tn_f32", Vt.data_ptr(), dMt.data_ptr(), dU.data_ptr(), P, Tp, cin, cout, _st(x))
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=dev)
    db = None
    if with_bias and cout % 4 == 0:             # as the gradient plan runs it: dW and db in one launch
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        _lib.call("bbdm_winograd_wgrad_finish_bias_f32", m, dU.data_ptr(), splits, dw.data_ptr(), cin, cout, dm11.data_ptr(), T,
                  db.data_ptr(), _st(x))
        return dw, db
    _lib.call("bbdm_winograd_wgrad_finish_f32", m, dU.data_ptr(), splits, dw.data_ptr(), cin, cout, _st(x))
    if with_bias:
        acc = torch.empty(cout, dtype=torch.float64, device=dev)
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        _lib.call("bbdm_colsum_f32", dm11.data_ptr(), cout, acc.data_ptr(), db.data_ptr(), T, cout, _st(x))
    return dw, db


def gemm_tn_batched(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a: [batch, K, M], b: [batch, K, N] -> sum over the K splits of C[z][batch][M][N] = a^T b (fp32 MFMA)."""
    _chk(a, b)
    batch, K, M = a.shape
    N = b.shape[2]
    lib = _lib.load()
    splits = lib.bbdm_gemm_tn_splits(batch, K, M, N)
    c = torch.empty(splits, batch, M, N, dtype=torch.float32, device=a.device)
    _lib.call("bbdm_gemm_tn_batched_f32", a.data_ptr(), M, K * M, b.data_ptr(), N, K * N, c.data_ptr(), batch, K, M, N, _st(a))
    out = c[0].clone()
    for z in range(1, splits):
        out += c[z]
    return out


def colsum(dy: torch.Tensor, C: Optional[int] = None) -> torch.Tensor:
    _chk(dy)
    ld = dy.shape[-1]
    C = C or ld
    M = dy.numel() // ld
    acc = torch.empty(4 * C, dtype=torch.float64, device=dy.device)       # limb cells
    out = torch.empty(C, dtype=torch.float32, device=dy.device)
    _lib.call("bbdm_colsum_f32", dy.data_ptr(), ld, acc.data_ptr(), out.data_ptr(), M, C, _st(dy))
    return out


def groupnorm_bwd(x, stats, gamma, beta, da, film=None, dadd=None, eps=1e-5, silu=False, resample=0, groups=32,
                  dx=None, accumulate=False):
    """Returns (dx, dgamma, dbeta, dfilm[N, 2C] or None)."""
    _chk(x, stats, gamma, beta, da, film, dadd, dx)
    N, H, W, C = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
    dfilm = torch.empty(N, 2 * C, dtype=torch.float32, device=x.device) if film is not None else None
    lib = _lib.load()
    ws = torch.empty(lib.bbdm_groupnorm_bwd_workspace_doubles(N, C, groups), dtype=torch.float64, device=x.device)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.call("bbdm_groupnorm_bwd_f32", x.data_ptr(), C, p(stats), p(gamma), p(beta), p(film),
              0 if film is None else film.shape[1], p(da), 0 if da is None else da.shape[-1], p(dadd),
              0 if dadd is None else dadd.shape[-1], dx.data_ptr(), dx.shape[-1], 1 if accumulate else 0, p(dgamma),
              p(dbeta), p(dfilm), 0 if dfilm is None else 2 * C, ws.data_ptr(), N, H, W, C, groups, float(eps),
              1 if silu else 0, resample, _st(x))
    return dx, dgamma, dbeta, dfilm


def linear_bwd(dy, x, w, act_in=False, need_dx=True, need_db=True):
    """Backward of :func:`linear` (x = its pre-activation input).  Returns (dx or None, dw, db or None)."""
    _chk(dy, x, w)
    N, In = x.shape
    Out = w.shape[0]
    lib = _lib.load()
    dw = torch.empty_like(w)
    db = torch.empty(Out, dtype=torch.float32, device=x.device) if need_db else None
    dx = torch.empty_like(x) if need_dx else None
    ws = torch.empty(max(1, lib.bbdm_linear_bwd_workspace_floats(min(N, 64), In, Out)), dtype=torch.float32, device=x.device)
    if N > 64:
        raise ValueError("linear_bwd: N <= 64 rows per call")
    _lib.call("bbdm_linear_bwd_f32", dy.data_ptr(), x.data_ptr(), w.data_ptr(), None if dx is None else dx.data_ptr(),
              dw.data_ptr(), None if db is None else db.data_ptr(), ws.data_ptr(), N, In, Out, 1 if act_in else 0, _st(x))
    return dx, dw, db


def gemm_bf3(V: torch.Tensor, w_packed_f32: torch.Tensor, batch: int, cin_pad: int, cout: int) -> torch.Tensor:
    """V: [batch, T, cin_pad] fp32; w_packed_f32: the fp32 packed weights ([batch][cin_pad/16][CoutPad][16]).  Returns
    M [batch, T, cout] computed by the bf16x3 kernel (csrc/gemm_bf3.hip)."""
    _chk(V, w_packed_f32)
    T = V.shape[1]
    lib = _lib.load()
    nh = lib.bbdm_gemm_bf3_packed_halfs(batch, cin_pad, cout)
    pk = torch.empty(nh, dtype=torch.int16, device=V.device)
    _lib.call("bbdm_gemm_bf3_pack_f32", w_packed_f32.data_ptr(), pk.data_ptr(), batch, cin_pad, cout, _st(V))
    M = torch.empty(batch, T, cout, dtype=torch.float32, device=V.device)
    _lib.call("bbdm_gemm_bf3_f32", V.data_ptr(), pk.data_ptr(), M.data_ptr(), batch, T, cin_pad, cout, _st(V))
    return M


def gemm_bf3p(V: torch.Tensor, w_packed_f32: torch.Tensor, batch: int, cin_pad: int, cout: int,
              bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The GEMM of :func:`gemm_bf3` on the pre-split kernel (csrc/gemm_bf3p.hip): V [batch, T, cin_pad] fp32 is split into
    its three bf16 planes by bbdm_gemm_bf3p_split_rows_f32, the weights by bbdm_gemm_bf3p_pack_b_f32.  T need not be a
    multiple of 256 (the planes are zero-padded; the result is cut back)."""
    _chk(V, w_packed_f32, bias, residual)
    T = V.shape[1]
    Tp = (T + 255) // 256 * 256
    lib = _lib.load()
    ap = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(batch, T, cin_pad), dtype=torch.uint8, device=V.device)
    bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(batch, cin_pad, cout), dtype=torch.uint8, device=V.device)
    _lib.call("bbdm_gemm_bf3p_split_rows_f32", V.data_ptr(), V.shape[2], ap.data_ptr(), batch, T, cin_pad, _st(V))
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", w_packed_f32.data_ptr(), bp.data_ptr(), batch, cin_pad, cout, _st(V))
    M = torch.empty(batch, Tp, cout, dtype=torch.float32, device=V.device)
    if residual is not None:
        M[:, :T] = residual
    _lib.call("bbdm_gemm_bf3p_f32", ap.data_ptr(), bp.data_ptr(), None if bias is None else bias.data_ptr(),
              None if residual is None else M.data_ptr(), cout, M.data_ptr(), cout, batch, Tp, cin_pad, cout, _st(V))
    return M[:, :T]


def gemm_bf3p_splitk(V: torch.Tensor, w_packed_f32: torch.Tensor, batch: int, cin_pad: int, cout: int, rows: int,
                     splits: int, fill: float = 0.0) -> torch.Tensor:
    """bbdm_gemm_bf3p_splitk_f32: only ``rows`` (a multiple of 32) of the T rows are computed, split z writes its partial sums to
    M[z] -> [splits, batch, Tp, cout] (rows >= ``rows`` keep ``fill``)."""
    _chk(V, w_packed_f32)
    T = V.shape[1]
    Tp = (T + 255) // 256 * 256
    lib = _lib.load()
    ap = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(batch, T, cin_pad), dtype=torch.uint8, device=V.device)
    bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(batch, cin_pad, cout), dtype=torch.uint8, device=V.device)
    _lib.call("bbdm_gemm_bf3p_split_rows_f32", V.data_ptr(), V.shape[2], ap.data_ptr(), batch, T, cin_pad, _st(V))
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", w_packed_f32.data_ptr(), bp.data_ptr(), batch, cin_pad, cout, _st(V))
    M = torch.full((splits, batch, Tp, cout), fill, dtype=torch.float32, device=V.device)
    _lib.call("bbdm_gemm_bf3p_splitk_f32", ap.data_ptr(), bp.data_ptr(), M.data_ptr(), cout, batch, Tp, rows, cin_pad, cout, splits,
              _st(V))
    return M


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """x: [rows, C] -> nn.LayerNorm(C) per row."""
    _chk(x, gamma, beta)
    y = torch.empty_like(x)
    _lib.call("bbdm_layernorm_f32", x.data_ptr(), x.shape[1], gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.shape[1],
              x.shape[0], x.shape[1], float(eps), _st(x))
    return y


def geglu(a: torch.Tensor) -> torch.Tensor:
    """a: [rows, 2 * inner] -> a[:, :inner] * gelu(a[:, inner:])."""
    _chk(a)
    inner = a.shape[1] // 2
    y = torch.empty(a.shape[0], inner, dtype=torch.float32, device=a.device)
    _lib.call("bbdm_geglu_f32", a.data_ptr(), a.shape[1], y.data_ptr(), inner, a.shape[0], inner, _st(a))
    return y


def cross_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, return_lse: bool = False):
    """q: [N, Tq, heads*ch], k / v: [N, Tk, heads*ch] (separate contiguous tensors of equal pitch)."""
    _chk(q, k, v)
    N, Tq, C = q.shape
    out = torch.empty_like(q)
    lse = torch.empty(N, heads, Tq, dtype=torch.float32, device=q.device) if return_lse else None
    _lib.call("bbdm_cross_attention_f32", q.data_ptr(), C, k.data_ptr(), v.data_ptr(), C, out.data_ptr(), C,
              None if lse is None else lse.data_ptr(), N, Tq, k.shape[1], heads, C // heads, _st(q))
    return (out, lse) if return_lse else out


def cross_attention_bwd(q, k, v, out, lse, dout, heads: int):
    """-> (dq, dk, dv) of cross_attention."""
    _chk(q, k, v, out, lse, dout)
    N, Tq, C = q.shape
    Tk = k.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    dwork = torch.empty(N * heads * Tq, dtype=torch.float32, device=q.device)
    _lib.call("bbdm_cross_attention_bwd_f32", q.data_ptr(), C, k.data_ptr(), v.data_ptr(), C, out.data_ptr(), C, dout.data_ptr(),
              C, lse.data_ptr(), dwork.data_ptr(), dq.data_ptr(), C, dk.data_ptr(), dv.data_ptr(), C, N, Tq, Tk, heads,
              C // heads, _st(q))
    return dq, dk, dv


def layernorm_bwd(x, gamma, dy, eps: float = 1e-5, dadd=None):
    """x, dy: [rows, C] -> (dx [+ dadd], dgamma, dbeta)."""
    _chk(x, gamma, dy, dadd)
    rows, C = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(8 * C, dtype=torch.float64, device=x.device)       # [2][C] limb cells
    _lib.call("bbdm_layernorm_bwd_f32", x.data_ptr(), C, gamma.data_ptr(), dy.data_ptr(), C,
              None if dadd is None else dadd.data_ptr(), C, dx.data_ptr(), C, dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, C,
              eps, _st(x))
    return dx, dg, db


def geglu_bwd(a, dy):
    """a: [rows, 2 * inner], dy: [rows, inner] -> da [rows, 2 * inner]."""
    _chk(a, dy)
    inner = a.shape[1] // 2
    da = torch.empty_like(a)
    _lib.call("bbdm_geglu_bwd_f32", a.data_ptr(), a.shape[1], dy.data_ptr(), inner, da.data_ptr(), a.shape[1], a.shape[0], inner,
              _st(a))
    return da


def vq_nearest(z: torch.Tensor, codebook: torch.Tensor):
    """z: [pixels, e_dim], codebook: [n_e, e_dim] -> (int64 indices [pixels], nearest rows [pixels, e_dim])."""
    _chk(z, codebook)
    P, D = z.shape
    idx = torch.empty(P, dtype=torch.int64, device=z.device)
    zq = torch.empty(P, D, dtype=torch.float32, device=z.device)
    _lib.call("bbdm_vq_nearest_f32", z.data_ptr(), D, codebook.data_ptr(), idx.data_ptr(), zq.data_ptr(), D, P,
              codebook.shape[0], D, _st(z))
    return idx, zq


def conv1x1_bf3q(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, cin: Optional[int] = None, x_off: int = 0, small: bool = False) -> torch.Tensor:
    """:func:`conv1x1_bf3` on the pipelined kernel with an fp32 A operand (bbdm_conv1x1_bf3q_f32; weights as bf3p B planes);
    ``small``: on the small-problem kernel (bbdm_conv1x1_bf3s_f32: same arguments, cin a multiple of 64)."""
    _chk(x, w, bias, residual)
    cout = w.shape[0]
    cin = cin or w.shape[1]
    lib = _lib.load()
    pf = torch.empty(lib.bbdm_conv_packed_floats(cout, cin, 1), dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), pf.data_ptr(), cout, w.shape[1], cin, 1, _st(x))
    bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(1, cin, cout), dtype=torch.uint8, device=x.device)
    _lib.call("bbdm_gemm_bf3p_pack_b_f32", pf.data_ptr(), bp.data_ptr(), 1, cin, cout, _st(x))
    if out is None:
        out = torch.empty(x.shape[0], cout, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv1x1_bf3s_f32" if small else "bbdm_conv1x1_bf3q_f32", x.data_ptr() + 4 * x_off, x.shape[1], bp.data_ptr(), None if bias is None else bias.data_ptr(),
              None if residual is None else residual.data_ptr(), 0 if residual is None else residual.shape[1],
              out.data_ptr(), out.shape[1], x.shape[0], cin, cout, _st(x))
    return out


def conv1x1_bf3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, cin: Optional[int] = None, x_off: int = 0) -> torch.Tensor:
    """x: [pixels, ldx] (channels x_off .. x_off + cin are convolved), w: [Cout, cin] -> [pixels, Cout] on the bf16x3 kernel."""
    _chk(x, w, bias, residual)
    cout = w.shape[0]
    cin = cin or w.shape[1]
    lib = _lib.load()
    pf = torch.empty(lib.bbdm_conv_packed_floats(cout, cin, 1), dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), pf.data_ptr(), cout, w.shape[1], cin, 1, _st(x))
    pk = torch.empty(lib.bbdm_gemm_bf3_packed_halfs(1, cin, cout), dtype=torch.int16, device=x.device)
    _lib.call("bbdm_gemm_bf3_pack_f32", pf.data_ptr(), pk.data_ptr(), 1, cin, cout, _st(x))
    if out is None:
        out = torch.empty(x.shape[0], cout, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv1x1_bf3_f32", x.data_ptr() + 4 * x_off, x.shape[1], pk.data_ptr(), None if bias is None else bias.data_ptr(),
              None if residual is None else residual.data_ptr(), 0 if residual is None else residual.shape[1],
              out.data_ptr(), out.shape[1], x.shape[0], cin, cout, _st(x))
    return out


def winograd_weight_planes(w: torch.Tensor, m: int, in_pad: int, dgrad: bool = False, fused: bool = True) -> torch.Tensor:
    """B planes of gemm_bf3p.hip for the Winograd-domain weights of ``w`` [Cout, Cin, 3, 3]: ``fused`` = one launch
    (bbdm_winograd_pack_weight_bf3p_f32), else bbdm_winograd_pack_weight_f32 + bbdm_gemm_bf3p_pack_b_f32."""
    _chk(w)
    lib = _lib.load()
    cout, cin = w.shape[0], w.shape[1]
    out_ch = cin if dgrad else cout
    planes = torch.zeros(lib.bbdm_gemm_bf3p_b_bytes((m + 2) ** 2, in_pad, out_ch), dtype=torch.uint8, device=w.device)
    if fused:
        _lib.call("bbdm_winograd_pack_weight_bf3p_f32", m, w.data_ptr(), planes.data_ptr(), cout, cin, in_pad, 1 if dgrad else 0, _st(w))
    else:
        f32 = torch.empty(lib.bbdm_winograd_packed_floats(m, out_ch, in_pad), dtype=torch.float32, device=w.device)
        _lib.call("bbdm_winograd_pack_weight_f32", m, w.data_ptr(), f32.data_ptr(), cout, cin, in_pad, 1 if dgrad else 0, _st(w))
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", f32.data_ptr(), planes.data_ptr(), (m + 2) ** 2, in_pad, out_ch, _st(w))
    return planes


# ---- the fp16-pair ("h2") planes of csrc/h2_split.h (round 6) -------------------------------------------------------------------------
def absmax(x: torch.Tensor) -> torch.Tensor:
    """A device float holding max |x| (bbdm_absmax_f32)."""
    _chk(x)
    b = torch.zeros(1, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_absmax_f32", x.data_ptr(), x.numel(), b.data_ptr(), _st(x))
    return b


def gemm_h2p(V: torch.Tensor, w_packed_f32: torch.Tensor, batch: int, cin_pad: int, cout: int,
             bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, bound_a: Optional[torch.Tensor] = None,
             bound_b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """:func:`gemm_bf3p` on two fp16 planes per operand: V [batch, T, cin_pad] fp32 is scaled by its bound (default: its own max) and split
    by bbdm_gemm_h2p_split_rows_f32, the weights by bbdm_gemm_h2p_pack_b_f32."""
    _chk(V, w_packed_f32, bias, residual)
    T = V.shape[1]
    Tp = (T + 255) // 256 * 256
    lib = _lib.load()
    ba = absmax(V) if bound_a is None else bound_a
    bb = absmax(w_packed_f32) if bound_b is None else bound_b
    ap = torch.empty(lib.bbdm_gemm_h2p_a_bytes(batch, T, cin_pad), dtype=torch.uint8, device=V.device)
    bp = torch.empty(lib.bbdm_gemm_h2p_b_bytes(batch, cin_pad, cout), dtype=torch.uint8, device=V.device)
    _lib.call("bbdm_gemm_h2p_split_rows_f32", V.data_ptr(), V.shape[2], ap.data_ptr(), ba.data_ptr(), batch, T, cin_pad, _st(V))
    _lib.call("bbdm_gemm_h2p_pack_b_f32", w_packed_f32.data_ptr(), bp.data_ptr(), bb.data_ptr(), 1.0, batch, cin_pad, cout, _st(V))
    M = torch.empty(batch, Tp, cout, dtype=torch.float32, device=V.device)
    if residual is not None:
        M[:, :T] = residual
    _lib.call("bbdm_gemm_h2p_f32", ap.data_ptr(), bp.data_ptr(), ba.data_ptr(), bb.data_ptr(), None if bias is None else bias.data_ptr(),
              None if residual is None else M.data_ptr(), cout, M.data_ptr(), cout, batch, Tp, cin_pad, cout, _st(V))
    return M[:, :T]


def conv3x3_winograd_planes(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], m: int, mode: str = "bf3",
                            pre_scale: Optional[torch.Tensor] = None, pre_bias: Optional[torch.Tensor] = None, pre_silu: bool = False,
                            in_bound: Optional[float] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 convolution as the product's inference plans stage it -- input transform writing planes, tile GEMMs on the pipe kernel, output
    transform -- with ``mode`` "bf3" (three bf16 planes) or "h2" (two fp16 planes; ``in_bound`` >= max |input of the transform|, default
    the measured maximum of x, for callers without a fused producer).  x [N, H, W, Cin] NHWC, w OIHW -> [N, H, W, Cout]."""
    _chk(x, w, bias, pre_scale, pre_bias, residual)
    lib = _lib.load()
    N, H, W, cin = x.shape
    cout = w.shape[0]
    planes = (m + 2) ** 2
    tiles = lib.bbdm_winograd_tiles(m, N, H, W)
    st = _st(x)
    pf = torch.empty(lib.bbdm_winograd_packed_floats(m, cout, cin), dtype=torch.float32, device=x.device)
    _lib.call("bbdm_winograd_pack_weight_f32", m, w.data_ptr(), pf.data_ptr(), cout, cin, cin, 0, st)
    Tp = (tiles + 255) // 256 * 256
    M = torch.empty(planes * Tp * cout, dtype=torch.float32, device=x.device)
    pre = (None if pre_scale is None else pre_scale.data_ptr(), None if pre_bias is None else pre_bias.data_ptr(),
           0 if pre_scale is None else pre_scale.shape[-1], 1 if pre_silu else 0)
    if mode == "h2":
        ub = absmax(w)                      # the filter's largest tap: packer and GEMM apply the gain of G . G^T themselves
        bp = torch.empty(lib.bbdm_gemm_h2p_b_bytes(planes, cin, cout), dtype=torch.uint8, device=x.device)
        _lib.call("bbdm_winograd_pack_weight_h2p_f32", m, w.data_ptr(), bp.data_ptr(), cout, cin, cin, 0, ub.data_ptr(), st)
        if in_bound is None:
            assert pre_scale is None, "a fused producer needs the caller's bound"
            vb = absmax(x)
        else:
            vb = torch.full((1,), float(in_bound), dtype=torch.float32, device=x.device)
        Vp = torch.empty(lib.bbdm_gemm_h2p_a_bytes(planes, tiles, cin), dtype=torch.uint8, device=x.device)
        _lib.call("bbdm_winograd_input_h2p_f32", m, x.data_ptr(), cin, Vp.data_ptr(), *pre, 0, N, H, W, cin, vb.data_ptr(), st)
        _lib.call("bbdm_winograd_gemm_h2p_f32", m, Vp.data_ptr(), bp.data_ptr(), M.data_ptr(), N, H, W, cin, cout, vb.data_ptr(),
                  ub.data_ptr(), st)
    else:
        bp = torch.empty(lib.bbdm_gemm_bf3p_b_bytes(planes, cin, cout), dtype=torch.uint8, device=x.device)
        _lib.call("bbdm_gemm_bf3p_pack_b_f32", pf.data_ptr(), bp.data_ptr(), planes, cin, cout, st)
        Vp = torch.empty(lib.bbdm_gemm_bf3p_a_bytes(planes, tiles, cin), dtype=torch.uint8, device=x.device)
        _lib.call("bbdm_winograd_input_bf3p_f32", m, x.data_ptr(), cin, Vp.data_ptr(), *pre, 0, N, H, W, cin, st)
        _lib.call("bbdm_winograd_gemm_bf3p_f32", m, Vp.data_ptr(), bp.data_ptr(), M.data_ptr(), N, H, W, cin, cout, st)
    out = torch.empty(N, H, W, cout, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_winograd_output_f32", m, M.data_ptr(), None if bias is None else bias.data_ptr(),
              None if residual is None else residual.data_ptr(), 0 if residual is None else residual.shape[-1], out.data_ptr(), cout, 0,
              N, H, W, cout, st)
    return out


def conv1x1_h2q(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, cin: Optional[int] = None, x_off: int = 0, xbound: Optional[torch.Tensor] = None,
                small: bool = False) -> torch.Tensor:
    """:func:`conv1x1_bf3q` on the fp16-pair planes (bbdm_conv1x1_h2q_f32): ``xbound`` a device float >= max |x| (default: the measured
    maximum of the whole buffer), the weights under their exact maximum."""
    _chk(x, w, bias, residual)
    cout = w.shape[0]
    cin = cin or w.shape[1]
    lib = _lib.load()
    pf = torch.empty(lib.bbdm_conv_packed_floats(cout, cin, 1), dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv_pack_weight_f32", w.data_ptr(), pf.data_ptr(), cout, w.shape[1], cin, 1, _st(x))
    wb = absmax(pf)
    bp = torch.empty(lib.bbdm_gemm_h2p_b_bytes(1, cin, cout), dtype=torch.uint8, device=x.device)
    _lib.call("bbdm_gemm_h2p_pack_b_f32", pf.data_ptr(), bp.data_ptr(), wb.data_ptr(), 1.0, 1, cin, cout, _st(x))
    xb = absmax(x) if xbound is None else xbound
    if out is None:
        out = torch.empty(x.shape[0], cout, dtype=torch.float32, device=x.device)
    _lib.call("bbdm_conv1x1_h2s_f32" if small else "bbdm_conv1x1_h2q_f32", x.data_ptr() + 4 * x_off, x.shape[1], bp.data_ptr(),
              None if bias is None else bias.data_ptr(),
              None if residual is None else residual.data_ptr(), 0 if residual is None else residual.shape[1],
              out.data_ptr(), out.shape[1], x.shape[0], cin, cout, xb.data_ptr(), wb.data_ptr(), _st(x))
    return out


def h2_rowl1(w: torch.Tensor, bias=None) -> torch.Tensor:
    """bbdm_h2_rowl1_f32: (max over rows of sum |w[row, :]| rounded up, max |bias|) as a two-element device tensor."""
    _chk(w)
    out = torch.full((2,), -1.0, dtype=torch.float32, device=w.device)
    _lib.call("bbdm_h2_rowl1_f32", w.data_ptr(), None if bias is None else bias.data_ptr(), w.shape[0], w[0].numel(), out.data_ptr(), _st(w))
    return out


def h2_affine_bound(in_bound: torch.Tensor, gain2: torch.Tensor) -> torch.Tensor:
    """bbdm_h2_affine_bound_f32: in_bound * gain2[0] + gain2[1] as a one-element device tensor."""
    out = torch.zeros(1, dtype=torch.float32, device=in_bound.device)
    _lib.call("bbdm_h2_affine_bound_f32", in_bound.data_ptr(), gain2.data_ptr(), out.data_ptr(), _st(in_bound))
    return out
