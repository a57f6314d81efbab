"""Training path: ``torch.autograd.Function`` wrappers over the HIP forward and backward plans.

``UNetModel.forward`` routes here when gradients are enabled.  The whole UNet is ONE autograd node: its forward runs
the training plan (same kernels as inference, activations kept), its backward runs the emitted gradient plan
(conv dgrad/wgrad on the fp32 matrix core, GroupNorm/FiLM/SiLU/resample, attention, embedding MLP -- DESIGN.md §4.4)
and returns a gradient for every parameter, so ``loss.backward()``, ``torch.optim.Adam`` and DDP's reducer hooks
(runners/BaseRunner.py:76,412-417) work as with the reference.  The reference's gradient checkpointing of the
attention block (util.py:119-148) has no numerical effect and is replaced by recomputing the attention
probabilities from the saved log-sum-exp.
"""
from __future__ import annotations

import torch

from . import _lib


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, t, context, *params):
        plan = model._plan_for(x, training=True)
        out = plan.run(x, t, context)
        ctx.plan, ctx.generation = plan, plan.generation
        ctx.cx = x.shape[1]
        ctx.cctx = 0 if context is None else context.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("bbdm_amd: the UNet was run again (same shape, training mode) before this backward; the "
                               "training plan keeps one set of saved activations per shape")
        need_x, need_ctx = ctx.needs_input_grad[1], ctx.needs_input_grad[3]
        flat, dx_in = plan.run_backward(dout.contiguous().float(), need_x or need_ctx)
        grads = []
        for i, p in enumerate(plan.param_list):
            if ctx.needs_input_grad[4 + i]:
                off = plan.grad_off[id(p)]
                grads.append(flat[off:off + p.numel()].view_as(p))
            else:
                grads.append(None)
        dx = dctx = None
        if dx_in is not None:
            nchw = dx_in.permute(0, 3, 1, 2)
            if need_x:
                dx = nchw[:, :ctx.cx].contiguous()
            if need_ctx:
                dctx = nchw[:, ctx.cx:ctx.cx + ctx.cctx].contiguous()
        return (None, dx, None, dctx, *grads)


def unet_apply(model, x, timesteps, context):
    x, ctx = model._check_inputs(x, context)
    t = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
    return _UNetFn.apply(model, x, t, ctx, *model.parameters())


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, pred, loss_type):
        t, p = target.contiguous().float(), pred.contiguous().float()
        partial = torch.zeros(1, dtype=torch.float64, device=p.device)
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        with _lib.device_guard(p.device):
            _lib.call("bbdm_bb_loss_f32", t.data_ptr(), p.data_ptr(), partial.data_ptr(), out.data_ptr(), p.numel(),
                      loss_type, _lib.current_stream(p.device))
        ctx.save_for_backward(t, p)
        ctx.loss_type = loss_type
        return out[0]

    @staticmethod
    def backward(ctx, g):
        t, p = ctx.saved_tensors
        gs = g.reshape(1).contiguous().float()
        dp = torch.empty_like(p)
        with _lib.device_guard(p.device):
            _lib.call("bbdm_bb_loss_bwd_f32", p.data_ptr(), t.data_ptr(), gs.data_ptr(), dp.data_ptr(), p.numel(),
                      ctx.loss_type, _lib.current_stream(p.device))
        return None, dp, None


def bb_loss(target, pred, loss_type: str):
    """mean|target - pred| ('l1') or mean (target - pred)^2 ('l2'), differentiable w.r.t. ``pred``
    (BrownianBridgeModel.py:114-117)."""
    _lib.require_gpu(pred, target)
    return _LossFn.apply(target, pred, {"l1": 0, "l2": 1}[loss_type])
