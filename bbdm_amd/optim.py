"""Optimizer step + EMA update of the training loop as one HIP pass.

Drop-in mirrors of what the reference's runner builds around the model:

* :func:`get_optimizer`  -- ``runners/utils.py:48-51`` (``get_optimizer(optim_config, parameters)``): for ``'Adam'`` it
  returns :class:`FusedAdam`, a ``torch.optim.Optimizer`` with ``torch.optim.Adam``'s constructor, ``param_groups`` and
  ``state_dict()`` layout (``state[p] = {step, exp_avg, exp_avg_sq}``), so ``ReduceLROnPlateau`` (``BBDMRunner.py:61-66``)
  and the runner's optimizer checkpoints (``BaseRunner.py:128-151``) work unchanged; other optimizers are torch's own.
* :class:`EMA`           -- ``runners/base/EMA.py:4-43``: same methods (``register / reset_device / update /
  apply_shadow / restore``) and the same ``shadow`` / ``backup`` dicts keyed by parameter name (the runner checkpoints
  ``ema.shadow`` directly, ``BaseRunner.py:125,169``).

Both hand the C-ABI entry ``bbdm_adam_ema_step_f32`` a device table of raw (param, grad, exp_avg, exp_avg_sq, shadow)
chunk pointers: ONE launch updates all 248 tensors.  When the EMA update is due in the same iteration the runner can fuse
it into the optimizer's pass with ``optimizer.step(ema=ema, ema_with_decay=...)``; called separately
(`ema.update(net)`, the unmodified runner) it is its own single pass.  No CPU / PyTorch fallback: parameters must be fp32
GPU tensors.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib

__all__ = ["FusedAdam", "EMA", "get_optimizer"]


class _ChunkTable:
    """Device table of ``BbdmOptChunk`` entries, rebuilt only when one of the pointers it holds changes."""

    def __init__(self):
        self.key = None
        self.dev = None
        self.n = 0

    def get(self, rows, device):
        """rows: list of (param, grad|None, exp_avg|None, exp_avg_sq|None, shadow|None) tensors of equal numel."""
        key = tuple((p.data_ptr(), 0 if g is None else g.data_ptr(), 0 if m is None else m.data_ptr(),
                     0 if v is None else v.data_ptr(), 0 if s is None else s.data_ptr(), p.numel())
                    for p, g, m, v, s in rows)
        if key != self.key:
            ce = _lib.load().bbdm_opt_chunk_elems()
            ent = []
            for pp, gp, mp, vp, sp, n in key:
                for off in range(0, n, ce):
                    b = 4 * off
                    ent.append((pp + b, gp + b if gp else 0, mp + b if mp else 0, vp + b if vp else 0,
                                sp + b if sp else 0, min(ce, n - off)))
            host = torch.empty(len(ent), 6, dtype=torch.int64)
            for i, (a, b, c, d, e, n) in enumerate(ent):
                host[i, 0], host[i, 1], host[i, 2], host[i, 3], host[i, 4] = a, b, c, d, e
                host[i, 5] = n                        # int n + int pad: little-endian low word = n, high word = 0
            self.dev = host.to(device)
            self.key, self.n = key, len(ent)
        return self.dev, self.n


def _check_param(p: torch.Tensor):
    _lib.require_gpu(p)
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise TypeError("bbdm_amd.optim works on contiguous fp32 parameters (the reference trains in fp32)")


def _launch(device, table, n, do_adam, group, step, ema_mode, ema_decay):
    with _lib.device_guard(device):
        _lib.call("bbdm_adam_ema_step_f32", table.data_ptr(), n, int(do_adam), float(group["lr"]) if group else 0.0,
                  float(group["betas"][0]) if group else 0.0, float(group["betas"][1]) if group else 0.0,
                  float(group["eps"]) if group else 0.0, float(group["weight_decay"]) if group else 0.0,
                  int(step), int(ema_mode), float(ema_decay), _lib.current_stream(device))


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (no amsgrad / maximize) with the whole ``step()`` as
    one launch.  State layout = torch's: ``state[p]['step']`` (a float32 scalar tensor on the CPU, as torch keeps it for
    non-capturable Adam), ``'exp_avg'``, ``'exp_avg_sq'`` -- an optimizer checkpoint written by either loads into the
    other."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("bbdm_amd.optim.FusedAdam: amsgrad is not implemented (the reference never sets it)")
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError(f"invalid Adam hyper-parameters lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False, fused=None))
        self._tables: Dict[tuple, _ChunkTable] = {}

    @torch.no_grad()
    def step(self, closure=None, ema: Optional["EMA"] = None, ema_with_decay: bool = True):
        """One Adam step for every parameter that has a gradient.  ``ema`` (optional): also apply that EMA's update for
        these parameters in the same pass (``EMA.update(net, with_decay=ema_with_decay)`` semantics, on the updated
        weights)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            kinds: Dict[tuple, list] = {}        # (device, step count) -> rows; normally ONE kind = one launch
            no_grad_rows = []                    # parameters without a gradient: no Adam update (as torch) -- but EMA.update covers
            for p in group["params"]:           # EVERY registered parameter (EMA.py:21-29), so the fused pass does too
                if p.grad is None:
                    shadow = ema._shadow_of(p) if ema is not None else None
                    if shadow is not None:
                        _check_param(p)
                        no_grad_rows.append((p, None, None, None, shadow))
                    continue
                _check_param(p)
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32:
                    raise RuntimeError("bbdm_amd.optim.FusedAdam needs dense fp32 gradients")
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                shadow = ema._shadow_of(p) if ema is not None else None
                kinds.setdefault((p.device, int(st["step"])), []).append((p, g, st["exp_avg"], st["exp_avg_sq"], shadow))
            mode = 0 if ema is None else (1 if ema_with_decay else 2)
            by_dev: Dict[torch.device, list] = {}
            for r in no_grad_rows:
                by_dev.setdefault(r[0].device, []).append(r)
            for device, rows in by_dev.items():
                table, n = self._tables.setdefault((gi, device, "ema-only"), _ChunkTable()).get(rows, device)
                _launch(device, table, n, False, None, 0, mode, ema.ema_decay)
            for (device, step_no), rows in kinds.items():
                table, n = self._tables.setdefault((gi, device, len(kinds) > 1 and step_no), _ChunkTable()).get(rows, device)
                _launch(device, table, n, True, group, step_no, mode, ema.ema_decay if ema is not None else 0.0)
                # the kernel rewrote the parameters behind autograd's back: bump their version counters, as an in-place torch op
                # would -- the UNet keys its packed weight copies (and autograd its saved-tensor checks) on them.  Without this the
                # next forward ran on the conv weights of BEFORE the step.
                torch.autograd.graph.increment_version([r[0] for r in rows])
        return loss


class EMA:
    """``runners/base/EMA.py`` with the update as one launch over all parameters."""

    def __init__(self, ema_decay):
        super().__init__()
        self.ema_decay = ema_decay
        self.backup = {}
        self.shadow = {}
        self._table = _ChunkTable()
        self._by_param: Dict[int, str] = {}

    def register(self, current_model: nn.Module):
        for name, param in current_model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()
                self._by_param[id(param)] = name

    def reset_device(self, current_model: nn.Module):
        for name, param in current_model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = self.shadow[name].to(param.data.device)
                self._by_param[id(param)] = name

    def _shadow_of(self, param) -> Optional[torch.Tensor]:
        name = self._by_param.get(id(param))
        return None if name is None else self.shadow.get(name)

    @torch.no_grad()
    def update(self, current_model: nn.Module, with_decay=True):
        rows, device = [], None
        for name, param in current_model.named_parameters():
            if param.requires_grad:
                assert name in self.shadow
                _check_param(param)
                sh = self.shadow[name]
                if sh.data_ptr() == param.data_ptr():
                    raise RuntimeError("EMA.update() between apply_shadow() and restore(): the weights ARE the shadow")
                if sh.device != param.device or sh.dtype != torch.float32 or not sh.is_contiguous():
                    sh = self.shadow[name] = sh.to(device=param.device, dtype=torch.float32).contiguous()
                self._by_param[id(param)] = name
                device = param.device
                rows.append((param, None, None, None, sh))
        if not rows:
            return
        table, n = self._table.get(rows, device)
        _launch(device, table, n, False, None, 0, 1 if with_decay else 2, self.ema_decay)

    def apply_shadow(self, current_model: nn.Module):
        for name, param in current_model.named_parameters():
            if param.requires_grad:
                assert name in self.shadow
                self.backup[name] = param.data
                param.data = self.shadow[name]

    def restore(self, current_model: nn.Module):
        for name, param in current_model.named_parameters():
            if param.requires_grad:
                assert name in self.backup
                param.data = self.backup[name]
        self.backup = {}


def get_optimizer(optim_config, parameters):
    """runners/utils.py:48-57 with Adam -> :class:`FusedAdam`."""
    if optim_config.optimizer == 'Adam':
        return FusedAdam(parameters, lr=optim_config.lr, weight_decay=optim_config.weight_decay,
                         betas=(optim_config.beta1, 0.999))
    elif optim_config.optimizer == 'RMSProp':
        return torch.optim.RMSprop(parameters, lr=optim_config.lr, weight_decay=optim_config.weight_decay)
    elif optim_config.optimizer == 'SGD':
        return torch.optim.SGD(parameters, lr=optim_config.lr, momentum=0.9)
    else:
        return NotImplementedError('Optimizer {} not understood.'.format(optim_config.optimizer))
