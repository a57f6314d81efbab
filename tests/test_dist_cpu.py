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
