"""The real model under DistributedDataParallel over RCCL, world size 2, one process per GPU (main.py:68-86,100-104 spawn the ranks;
runners/BaseRunner.py:76 wraps the model).  Needs >= 2 GPUs: skipped on the single-GPU boxes of this pool (the same body runs over
gloo on the CPU-emulated kernels in tests/test_dist_cpu.py); written so that the first multi-GPU node exercises RCCL with N > 1.

Checks: the chain-of-segments autograd graph drives DDP's reducer (several 1 MB buckets), accumulation_sync() skips the collective on
the non-boundary micro-step, FusedAdam steps, both ranks end with identical parameters, and the reduced gradient equals autograd on
the oracle over the COMBINED batch -- for a concat-conditioned UNet and for one with SpatialTransformer blocks (whose channel-padded
to_k / to_v weight gradients leave their backward segment through a scratch tensor)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, case):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    from bbdm_amd import dist_utils as du
    from bbdm_amd.optim import FusedAdam
    from fixtures import load_case
    import test_dist_cpu as DC
    import test_training_gpu as T
    dist = du.init(backend="nccl")
    info = du.gather_device_info(dist, dev)
    rec = load_case(case)
    m = T.build(rec, dev).train()
    m.denoise_fn.hip_graph = False
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[rank], output_device=rank, bucket_cap_mb=1)
    opt = FusedAdam(m.get_parameters(), lr=1e-4)
    x0, y, t, nz = (rec[k] for k in ("x0", "y", "t", "noise"))
    n = x0.shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    losses = []
    for step in (1, 2):                                       # two micro-steps, ONE all-reduce (on the boundary step)
        with du.accumulation_sync(ddp, step, 2):
            loss, _ = DC._ddp_losses(ddp, x0[sl].to(dev), y[sl].to(dev), t[sl].to(dev), nz[sl].to(dev))
            loss.backward()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize(dev)
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    opt.step()
    torch.cuda.synchronize(dev)
    q.put((rank, losses, {k: v.numpy().tolist() for k, v in grads.items()
                          if k.endswith("out.2.bias") or k.endswith("time_embed.0.bias") or k.endswith("attn2.to_k.weight")
                          or k.endswith("input_blocks.0.0.weight")},
           [float(p.detach().double().sum()) for p in m.get_parameters()], info))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL with N > 1)")
@pytest.mark.timeout(600)
@pytest.mark.parametrize("case,port", [("tiny_concat", 29631), ("tiny_xattn", 29633)])
def test_two_rank_rccl_ddp_training_step(case, port):
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=540) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, g0, sums0, info0), (_, l1, g1, sums1, _) = res
    assert sums0 == sums1                                           # identical parameters after the step
    assert g0 == g1                                                 # identical (reduced) gradients
    assert len(info0) == world and {d["device"] for d in info0} == {"cuda:0", "cuda:1"} and "rccl_version" in info0[0]
    from fixtures import load_case
    import test_training_gpu as T
    rec = load_case(case)
    used = (rec["x0"].shape[0] // world) * world
    for k in ("x0", "y", "t", "noise"):
        rec[k] = rec[k][:used]
    _, g_ref = T._oracle_grads(rec)
    for k, v in g0.items():
        ref = 2.0 * g_ref[k]             # two accumulated micro-steps on the same data = 2 x the gradient of the mean loss
        got = torch.tensor(v)
        assert float((got - ref).abs().max()) < 1e-3 * max(float(ref.abs().max()), 1e-6), k
