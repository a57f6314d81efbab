"""bbdm_amd.optim on the GPU: FusedAdam against torch.optim.Adam (HIP), EMA against the reference's EMA
arithmetic, to 1e-6 of the parameter magnitude; and the whole-model pass at the real 237 M-parameter size with its
achieved HBM rate printed."""
import pytest
import torch

import optim_cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("wd,beta1", [(0.0, 0.9), (0.01, 0.5)])
def test_fused_adam_matches_torch_adam(dev, wd, beta1):
    C.adam_parity(dev, wd, beta1)


def test_ema_matches_reference_ema_and_fuses_into_the_step(dev):
    C.ema_parity(dev)


def test_full_size_step_one_launch_and_rate(dev):
    """All 248 tensors / 237 M parameters of the Template UNet: one launch, results equal to torch.optim.Adam's foreach path
    + the EMA formula; prints the achieved HBM rate of the fused pass (9 x 4 B per parameter with the EMA)."""
    import bench
    import bbdm_amd
    from bbdm_amd.optim import EMA, FusedAdam
    up = bench.WORKLOADS["c4"][1]
    net = bbdm_amd.unet.UNetModel(**up).to(dev)
    ref = bbdm_amd.unet.UNetModel(**up).to(dev)
    ref.load_state_dict(net.state_dict())
    g = torch.Generator(device=dev).manual_seed(3)
    for p, q in zip(net.parameters(), ref.parameters()):
        p.data.normal_(0, 0.02, generator=g)
        q.data.copy_(p.data)
        p.grad = torch.randn(p.shape, device=dev, generator=g) * 1e-3
        q.grad = p.grad.clone()
    ema = EMA(0.995)
    ema.register(net)
    shadow_ref = {k: v.clone() for k, v in ema.shadow.items()}
    opt, opt_ref = FusedAdam(net.parameters(), lr=1e-4), torch.optim.Adam(ref.parameters(), lr=1e-4)
    for it in range(3):
        opt.step(ema=ema, ema_with_decay=True)
        opt_ref.step()
        for k, q in ref.named_parameters():
            shadow_ref[k] = (1.0 - 0.995) * q.data + 0.995 * shadow_ref[k]
    torch.cuda.synchronize()
    worst = 0.0
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        worst = max(worst, C.rel(p.data, q.data), C.rel(ema.shadow[k], shadow_ref[k]))
    n = sum(p.numel() for p in net.parameters())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        opt.step(ema=ema, ema_with_decay=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        opt_ref.step()
    t1.record()
    torch.cuda.synchronize()
    print(f"fused Adam+EMA over {n / 1e6:.1f} M parameters: {ms:.3f} ms/step = {9 * 4 * n / ms / 1e9:.2f} TB/s "
          f"(torch.optim.Adam alone: {t0.elapsed_time(t1) / 5:.3f} ms); worst rel err {worst:.2e}")
    assert worst < 1e-6
