"""Shared bodies of the optimizer / EMA parity tests (run on the GPU by test_optim_gpu.py and on the CPU-emulated kernels
by test_optim_emu_cpu.py): bbdm_amd.optim against torch.optim.Adam and the reference's own EMA class."""
import os
import sys

import pytest
import torch
import torch.nn as nn


def reference_ema_class():
    if os.path.isdir("/root/reference/runners/base"):
        sys.path.insert(0, "/root/reference")
        from runners.base.EMA import EMA
        return EMA

    class EMA:                                   # runners/base/EMA.py:4-29
        def __init__(self, ema_decay):
            self.ema_decay, self.backup, self.shadow = ema_decay, {}, {}

        def register(self, model):
            for name, param in model.named_parameters():
                if param.requires_grad:
                    self.shadow[name] = param.data.clone()

        def update(self, model, with_decay=True):
            for name, param in model.named_parameters():
                if param.requires_grad:
                    if with_decay:
                        new_average = (1.0 - self.ema_decay) * param.data + self.ema_decay * self.shadow[name]
                    else:
                        new_average = param.data
                    self.shadow[name] = new_average.clone()
    return EMA


def make_net(seed, dev=None):
    torch.manual_seed(seed)
    # odd sizes on purpose: unaligned chunk starts, tails, a tensor larger than one 16384-element chunk, a 3-element bias
    return nn.Sequential(nn.Conv2d(5, 37, 3), nn.GroupNorm(1, 37), nn.Linear(37, 3), nn.Conv2d(37, 150, 3)).to(dev)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def adam_parity(dev, wd, beta1):
    from bbdm_amd.optim import FusedAdam
    a, b = make_net(1, dev), make_net(1, dev)
    oa = FusedAdam(a.parameters(), lr=1e-3, betas=(beta1, 0.999), weight_decay=wd)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, betas=(beta1, 0.999), weight_decay=wd)
    # weight decay: ATen forms g + wd * p with one rounding on the CPU (what the kernel does: fmaf) and with two in its HIP
    # foreach kernels; where g and wd * p cancel that last-ulp difference is amplified by 1 / (|g'| + eps)
    tol = 1e-6 if wd == 0.0 else 3e-6
    g = torch.Generator().manual_seed(5)
    for it in range(6):
        for pa, pb in zip(a.parameters(), b.parameters()):
            gr = (torch.randn(pa.shape, generator=g) * (10.0 ** (it - 3))).to(dev)          # spans 1e-3 .. 1e2
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if it == 3:                                   # a parameter without a gradient is skipped by both
            a[2].bias.grad = b[2].bias.grad = None
        if it == 4:                                   # lr changed by a scheduler between steps
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-4
        oa.step()
        ob.step()
        for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert rel(pa, pb) < tol, (it, k, rel(pa, pb))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for i in sa["state"]:
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"])
        assert rel(sa["state"][i]["exp_avg"], sb["state"][i]["exp_avg"]) < 1e-6
        assert rel(sa["state"][i]["exp_avg_sq"], sb["state"][i]["exp_avg_sq"]) < 1e-6
    # an optimizer checkpoint of torch's Adam loads into FusedAdam and vice versa
    oa.load_state_dict(sb)
    ob.load_state_dict(sa)


def ema_parity(dev):
    from bbdm_amd.optim import EMA, FusedAdam
    RefEMA = reference_ema_class()
    a, b, c = make_net(2, dev), make_net(2, dev), make_net(2, dev)
    ea, eb, ec = EMA(0.995), RefEMA(0.995), EMA(0.995)
    ea.register(a); eb.register(b); ec.register(c)
    oa = FusedAdam(a.parameters(), lr=1e-3)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3)
    oc = FusedAdam(c.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(9)
    for it in range(5):
        for pa, pb, pc in zip(a.parameters(), b.parameters(), c.parameters()):
            gr = torch.randn(pa.shape, generator=g).to(dev)
            pa.grad, pb.grad, pc.grad = gr.clone(), gr.clone(), gr.clone()
        if it == 3:                                   # a parameter without a gradient: no Adam update, but EMA.update covers EVERY
            a[2].bias.grad = b[2].bias.grad = c[2].bias.grad = None      # registered parameter (EMA.py:21-29) -- the fused pass too
        decay = it >= 2                               # with_decay=False before start_ema_step (BaseRunner.py:174)
        oa.step(); ea.update(a, with_decay=decay)     # the unmodified runner's order: step, then EMA.update
        ob.step(); eb.update(b, with_decay=decay)
        oc.step(ema=ec, ema_with_decay=decay)         # one pass
        for k in eb.shadow:
            assert rel(ea.shadow[k], eb.shadow[k]) < 1e-6, (it, k)
            assert rel(ec.shadow[k], eb.shadow[k]) < 1e-6, (it, k)
    # apply_shadow / restore swap param.data exactly like the reference (EMA.py:31-43)
    before = {k: p.data_ptr() for k, p in a.named_parameters()}
    ea.apply_shadow(a)
    assert all(p.data_ptr() == ea.shadow[k].data_ptr() for k, p in a.named_parameters())
    with pytest.raises(RuntimeError):
        ea.update(a)
    ea.restore(a)
    assert {k: p.data_ptr() for k, p in a.named_parameters()} == before and ea.backup == {}
