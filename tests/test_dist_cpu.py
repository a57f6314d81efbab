"""N > 1 path on CPU: world_size-2 gloo processes exercising the sharding / timing / aggregation helpers that
bench.py and the sampling driver use (the GPU compute itself cannot run here: no CPU fallback by design)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from bbdm_amd import dist_utils as du
    dist = du.init(backend="gloo")
    assert dist is not None and dist.get_world_size() == world
    total = 37
    b, e = du.shard_range(total, rank, world)
    # every rank "samples" its shard: here a deterministic function of the unit index
    mine = torch.zeros(total)
    mine[b:e] = torch.arange(b, e, dtype=torch.float32) * 2 + 1
    import time
    elapsed = du.timed_region(lambda: time.sleep(0.05 * (rank + 1)), dist, None)
    dist.all_reduce(mine)                      # control-plane check only: shards are disjoint and exhaustive
    # DDP-style gradient averaging over gloo on a tiny module (what DDP does for the UNet over RCCL)
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    ddp = torch.nn.parallel.DistributedDataParallel(lin)
    x = torch.full((2, 4), float(rank + 1))
    ddp(x).sum().backward()
    q.put((rank, b, e, elapsed, mine.tolist(), lin.weight.grad.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharding_timing_and_grad_allreduce():
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, b0, e0, t0, m0, g0), (r1, b1, e1, t1, m1, g1) = res
    assert (b0, e0, b1, e1) == (0, 19, 19, 37)                      # disjoint, exhaustive, sizes differ by <= 1
    assert m0 == m1 == [2.0 * i + 1 for i in range(37)]
    assert abs(t0 - t1) < 1e-9 and t0 >= 0.1                        # max over ranks, identical on both
    assert g0 == g1 == [[3.0] * 4] * 3                              # mean over ranks of (2*1, 2*2)


def test_shard_range_properties():
    from bbdm_amd.dist_utils import aggregate_throughput, shard_range
    for total in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1
    assert aggregate_throughput(10, 2.0, 8) == 40.0


# ---- the real model under DDP, world size 2, on the CPU-emulated kernels ---------------------------------------------
def _ddp_worker(rank, world, port, q, case="tiny_nocond"):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from bbdm_amd import dist_utils as du
    from bbdm_amd.optim import FusedAdam
    from emu_backend import emulated_backend
    from fixtures import load_case
    import test_training_gpu as T
    dist = du.init(backend="gloo")
    with emulated_backend():
        rec = load_case(case)
        m = T.build(rec, torch.device("cpu")).train()
        m.denoise_fn.hip_graph = False
        assert len(next(iter([m.denoise_fn._plan_for(rec["x0"][:1], True)])).bsegs) > 1      # a chain, not one node
        ddp = torch.nn.parallel.DistributedDataParallel(m, bucket_cap_mb=1)                  # several buckets
        opt = FusedAdam(m.get_parameters(), lr=1e-4)
        x0, y, t, nz = (rec[k] for k in ("x0", "y", "t", "noise"))
        n = x0.shape[0] // world
        sl = slice(rank * n, (rank + 1) * n)
        accumulate = 2
        hooks = []
        for step in (1, 2):                                       # two micro-steps, ONE all-reduce (on the boundary step)
            with du.accumulation_sync(ddp, step, accumulate):
                loss, _ = _ddp_losses(ddp, x0[sl], y[sl], t[sl], nz[sl])
                loss.backward()
            hooks.append(float(loss.detach()))
        grads = {k: p.grad.clone() for k, p in m.named_parameters()}
        opt.step()
        q.put((rank, hooks, {k: v.tolist() for k, v in list(grads.items())[:3]},
               [float(p.detach().double().sum()) for p in m.get_parameters()],
               {k: v.numpy().tolist() for k, v in grads.items() if k.endswith("out.2.bias") or k.endswith("time_embed.0.bias")
                or k.endswith("attn2.to_k.weight") or k.endswith("attn2.to_v.weight")}))
    dist.barrier()
    dist.destroy_process_group()


def _ddp_losses(ddp, x0, y, t, nz):
    """net(x, x_cond) with the timestep / noise pinned (BrownianBridgeModel.forward draws them at random)."""
    import bbdm_amd.model as M
    orig_randint, orig_randn = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t
    torch.randn_like = lambda *a, **k: nz
    try:
        return ddp(x0, y)
    finally:
        torch.randint, torch.randn_like = orig_randint, orig_randn


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case,port", [("tiny_nocond", 29617), ("tiny_xattn", 29619)])
def test_two_rank_ddp_training_step_on_the_emulated_kernels(case, port):
    """World size 2 over gloo: the chain-of-segments autograd graph drives DDP's reducer (buckets become ready segment by
    segment), accumulation_sync() skips the collective on the non-boundary micro-step, FusedAdam steps; both ranks end with
    identical parameters, and the reduced gradient equals autograd on the oracle over the COMBINED batch."""
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=540) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, _, sums0, g0), (_, l1, _, sums1, g1) = res
    assert sums0 == sums1                                           # identical parameters after the step
    assert g0 == g1                                                 # identical (reduced) gradients
    # oracle: two accumulated micro-steps on the same data = 2 x the gradient of the mean loss over the combined batch
    from fixtures import load_case
    import test_training_gpu as T
    rec = load_case(case)
    if case == "tiny_xattn":            # the padded to_k / to_v weight gradients left their segment filled (not as zero views)
        assert sum(k.endswith("attn2.to_k.weight") or k.endswith("attn2.to_v.weight") for k in g0) >= 2
    used = (rec["x0"].shape[0] // world) * world                   # each rank took x0.shape[0] // world samples
    for k in ("x0", "y", "t", "noise"):
        rec[k] = rec[k][:used]
    _, g_ref = T._oracle_grads(rec)
    for k, v in g0.items():
        ref = 2.0 * g_ref[k]
        got = torch.tensor(v)
        assert float((got - ref).abs().max()) < 1e-3 * max(float(ref.abs().max()), 1e-6), k
