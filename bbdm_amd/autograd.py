"""Training path: ``torch.autograd.Function`` wrappers over the HIP forward and backward plans.

``UNetModel.forward`` routes here when gradients are enabled.  The UNet is a short CHAIN of autograd nodes (one per
backward segment of the plan, ``unet._Plan.N_SEGMENTS``): the forward runs the training plan once (same kernels as
inference, activations kept); each node's backward runs its run of the emitted gradient plan (conv dgrad/wgrad on the fp32
matrix core, GroupNorm/FiLM/SiLU/resample, attention, embedding MLP -- DESIGN.md §4.4) and returns the gradients of the
parameters that run completed, so ``loss.backward()``, the optimizer and DDP's reducer hooks (runners/BaseRunner.py:76,
412-417) work as with the reference -- and DDP's bucketed all-reduce of the late layers overlaps the backward of the early
ones.  The reference's gradient checkpointing of the
attention block (util.py:119-148) has no numerical effect and is replaced by recomputing the attention
probabilities from the saved log-sum-exp.
"""
from __future__ import annotations

import torch

from . import _lib


class _UNetSeg(torch.autograd.Function):
    """One link of the UNet's autograd chain.  Link 0 (created first) runs the whole forward plan; links 1..K-1 only pass a
    token along; the last link returns the UNet output.  In backward the engine visits the links in reverse: link K-1 runs
    backward segment 0 (head side) of the plan and returns ITS parameters' gradients -- which the engine accumulates (DDP:
    reduces) at once -- then link K-2 runs segment 1, ... link 0 runs the last segment (+ the embedding path, + d input)."""

    @staticmethod
    def forward(ctx, model, plan, link, x, t, context, token, *params):
        K = len(plan.bsegs)
        ctx.plan, ctx.link, ctx.K = plan, link, K
        if link == 0:
            out = plan.run(x, t, context)
            plan._fwd_out = out
            ctx.cx = x.shape[1]
            ctx.cctx = 0 if context is None else context.shape[1]
        ctx.generation = plan.generation
        if link == K - 1:
            out, plan._fwd_out = plan._fwd_out, None
            return out
        return torch.zeros(1, dtype=torch.float32, device=plan.device)

    @staticmethod
    def backward(ctx, dout):
        plan, link, K = ctx.plan, ctx.link, ctx.K
        if plan.generation != ctx.generation:
            raise RuntimeError("bbdm_amd: the UNet was run again (same shape, training mode) before this backward; the "
                               "training plan keeps one set of saved activations per shape")
        seg = K - 1 - link
        if seg == 0:
            plan.backward_begin(dout.contiguous().float())
        need_x = link == 0 and ctx.needs_input_grad[3]
        need_ctx = link == 0 and ctx.needs_input_grad[5]
        dx_in = plan.backward_segment(seg, need_x or need_ctx)
        params = plan.bsegs[seg][2]
        if plan.m.grad_in_place and _accumulate_in_place(plan, params, [ctx.needs_input_grad[7 + i] for i in range(len(params))]):
            grads = [None] * len(params)            # already added to the .grad tensors
        else:
            grads = [plan.grad_view(p) if ctx.needs_input_grad[7 + i] else None for i, p in enumerate(params)]
        dx = dctx = None
        if dx_in is not None:
            nchw = dx_in.permute(0, 3, 1, 2)
            if need_x:
                dx = nchw[:, :ctx.cx].contiguous()
            if need_ctx:
                dctx = nchw[:, ctx.cx:ctx.cx + ctx.cctx].contiguous()
                via_attention = plan.context_token_grad()         # SpatialTransformer: + the cross-attention keys / values
                if via_attention is not None:
                    dctx = dctx + via_attention
        dtoken = None if link == 0 else torch.zeros(1, dtype=torch.float32, device=plan.device)
        return (None, None, None, dx, None, dctx, dtoken, *grads)


def _observed(q) -> bool:
    """Does anything registered ON THE PARAMETER watch gradients arrive?  Tensor hooks (``register_hook``) and post-accumulate-grad hooks
    (``register_post_accumulate_grad_hook``: optimizer-in-backward, FSDP2).  Hooks on the parameter's AccumulateGrad NODE (DDP's reducer,
    Horovod's DistributedOptimizer) cannot be seen from Python: dist_utils.accumulation_sync therefore only enables this path where it
    knows who owns the gradients (a bare model in a single-process job, torch's DDP under ``no_sync``), or where the caller says so."""
    return bool(q._backward_hooks or getattr(q, "_post_accumulate_grad_hooks", None))


def _accumulate_in_place(plan, params, needed) -> bool:
    """Gradient accumulation without autograd's per-parameter adds (UNetModel.grad_in_place, set by dist_utils.accumulation_sync).

    After the first micro-step of an accumulation cycle every ``p.grad`` is a view of ONE of the plan's two flat gradient buffers
    (autograd adopts the views :meth:`_Plan.grad_view` hands out), at the parameter's own offset; this micro-step's gradients lie at
    the same offsets of the OTHER flat buffer.  When that holds for every parameter of the segment, the segment's gradients are added
    with one ``add_`` per contiguous run of offsets (a handful per segment) and autograd receives None for them -- 248 ``AccumulateGrad``
    launches per micro-step become ~10.  Anything else (first micro-step, a foreign ``.grad``, a parameter with tensor hooks or
    post-accumulate-grad hooks -- optimizer-in-backward, FSDP2) returns False and the caller hands the views to autograd as before:
    returning None would silently skip those observers (see _observed for what cannot be detected and who guards against it)."""
    new = plan._flat_grad
    runs = []
    base = None
    for q, need in zip(params, needed):
        if not need:
            return False
        g = q.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != q.shape or _observed(q):
            return False
        off = plan.grad_off[id(q)]
        b = g.data_ptr() - 4 * off
        if base is None:
            base = b
        if b != base or b == new.data_ptr():
            return False
        runs.append((off, off + q.numel()))
    old = next((fb for fb in getattr(plan, "_flat_bufs", []) if fb.data_ptr() == base), None)
    if old is None or old.numel() != new.numel():
        return False
    runs.sort()
    merged = [list(runs[0])]
    for a, b in runs[1:]:
        if a == merged[-1][1]:
            merged[-1][1] = b
        else:
            merged.append([a, b])
    with torch.no_grad():
        for a, b in merged:
            old[a:b].add_(new[a:b])
    return True


def unet_apply(model, x, timesteps, context):
    x, ctx = model._check_inputs(x, context)
    t = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
    plan = model._plan_for(x, training=True)
    K = len(plan.bsegs)
    token = None
    for link in range(K):                       # link j owns the parameters of backward segment K-1-j
        params = plan.bsegs[K - 1 - link][2]
        token = _UNetSeg.apply(model, plan, link, x if link == 0 else None, t if link == 0 else None,
                               ctx if link == 0 else None, token, *params)
    return token


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, pred, loss_type):
        t, p = target.contiguous().float(), pred.contiguous().float()
        partial = torch.zeros(4, dtype=torch.float64, device=p.device)        # one exact limb cell (csrc/stats_acc.h)
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
