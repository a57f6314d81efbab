"""One-process-per-GPU helpers shared by bench.py and the sampling driver.

The path shards by image pair: every rank owns a contiguous slice of the batch / dataset and runs
the whole sampling loop on it; there is NO data-path collective.  The only exchanges are the control-plane ones the
reference also has (runners/BaseRunner.py:419,536: a scalar reduce for logging, a barrier per epoch) plus, for
training, DDP's gradient all-reduce which ``torch.nn.parallel.DistributedDataParallel`` issues itself.
Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` for the CPU tests.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional, Tuple

import torch


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun / torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None):
    """Initialise the default process group when WORLD_SIZE > 1; returns torch.distributed or None."""
    rank, _, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank,
                                world_size=world)
    return dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, disjoint, exhaustive [begin, end) split of ``total`` units over ``world`` ranks (sizes differ by
    at most one)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def timed_region(fn: Callable[[], None], dist=None, device: Optional[torch.device] = None) -> float:
    """Run ``fn`` bracketed by barrier + device synchronize on both sides; return the MAX elapsed seconds over ranks."""
    def fence():
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()

    fence()
    t0 = time.perf_counter()
    fn()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_throughput(units_per_rank: int, elapsed_max: float, world: int) -> float:
    """Whole-job units/s for weak scaling: every rank processed ``units_per_rank`` in the (max-over-ranks) time."""
    return world * units_per_rank / elapsed_max


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_procs: int, script_argv, python: Optional[str] = None) -> int:
    """Re-exec ``script_argv`` as ``n_procs`` ranks of one node under ``torch.distributed.run`` (one process per GPU;
    rendezvous on 127.0.0.1: the container hostname may not resolve).  The reference starts its ranks itself as well
    (main.py:100-104, ``mp.spawn``).  Returns the launcher's exit code."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n_procs))))
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_procs}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + list(script_argv)
    return subprocess.call(cmd, env=env)


def gather_device_info(dist=None, device: Optional[torch.device] = None):
    """[{rank, device, name}] for every rank (+ the RCCL version on GPUs), gathered to all ranks; control plane only."""
    rank, local_rank, world = env_rank()
    mine = {"rank": rank, "device": str(device) if device is not None else "cpu"}
    if device is not None and device.type == "cuda":
        mine["name"] = torch.cuda.get_device_name(device)
    if dist is None:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    if device is not None and device.type == "cuda":
        try:
            out[0]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                   # pragma: no cover
            pass
    return out


def _plain_modules():
    """The model classes of this package: bare (unwrapped) modules whose gradients nobody else manages."""
    from .model import BrownianBridgeModel
    from .unet import UNetModel
    return (BrownianBridgeModel, UNetModel)


def accumulation_sync(net, micro_step: int, accumulate_grad_batches: int, in_place=None):
    """Context manager for one micro-step of gradient accumulation under DDP.

    The reference all-reduces the full 948 MB gradient on EVERY micro-step (runners/BaseRunner.py:412-417: ``loss.backward()``
    with no ``no_sync()``) although the optimizer only steps every ``accumulate_grad_batches``-th one.  Skipping the
    collective on the non-boundary micro-steps (``net.no_sync()``) and reducing the locally accumulated sum on the boundary
    step gives the same averaged gradient -- a sum of means is the mean of sums -- with 1/accumulate_grad_batches of the
    RCCL traffic.  ``micro_step`` is the runner's 1-based ``global_step``; modules without ``no_sync`` get none.

    Inside the context the UNet may also add a micro-step's parameter gradients to the ``.grad`` tensors IN PLACE -- one fused add
    per backward segment instead of one ``AccumulateGrad`` add per parameter (248 launches per micro-step, bbdm_amd/autograd.py) --
    which BYPASSES autograd's AccumulateGrad nodes, so it is only enabled where it is known that nothing hangs on them
    (round-5 advisor finding: hooks on those nodes cannot be detected):
      * ``in_place`` = None (default): this package's own model, unwrapped, in a job with no process group of more than one rank and
        no Horovod loaded (nobody else can be reducing its gradients); and ``torch.nn.parallel.DistributedDataParallel`` on its
        non-boundary micro-steps (under ``no_sync`` its reducer ignores the hooks; on the boundary step they must fire, so that step
        takes autograd's path).  Any other wrapper (FSDP, a hand-written reducer ...) keeps autograd's accumulation on every micro-step.
      * ``in_place`` = True / False: the caller's word -- True only if nothing observes gradient arrival on the non-boundary micro-steps.
    Parameters with tensor hooks or post-accumulate-grad hooks are excluded in any case (bbdm_amd/autograd.py: _observed)."""
    import contextlib
    import sys
    from torch.nn.parallel import DistributedDataParallel
    boundary = accumulate_grad_batches <= 1 or micro_step % accumulate_grad_batches == 0
    ddp = isinstance(net, DistributedDataParallel)
    multi = (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1) \
        or "horovod.torch" in sys.modules
    bare = isinstance(net, _plain_modules()) and not multi      # this package's own model, unwrapped, and nobody to reduce with
    if in_place is None:
        allow = bare or (ddp and not boundary)
    else:
        allow = bool(in_place) and not (ddp and boundary)

    @contextlib.contextmanager
    def ctx():
        from .unet import UNetModel
        unets = [m for m in net.modules() if isinstance(m, UNetModel)] if allow else []
        for m in unets:
            m.grad_in_place = True
        try:
            if (ddp or hasattr(net, "no_sync")) and not boundary:
                with net.no_sync():
                    yield
            else:
                yield
        finally:
            for m in unets:
                m.grad_in_place = False
    return ctx()
