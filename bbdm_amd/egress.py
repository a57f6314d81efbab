"""Sample egress: generated batches -> PNG files without stalling the sampling loop.

The reference writes every image with ``save_single_image`` (``runners/utils.py:67-74``) from the main thread of
``BBDMRunner.sample_to_eval`` (``BBDMRunner.py:236-253``): per image a clone, five elementwise ATen kernels, a permute, a
blocking ``.to('cpu', torch.uint8)`` and a PNG encode -- 3 x batch_size times per batch, with the GPU idle meanwhile.

Here a whole batch is converted by ONE kernel (``bbdm_images_to_u8_f32``: the same fp32 arithmetic step by step, so the
bytes are identical), copied to pinned host memory with ONE asynchronous D2H copy, and encoded by a small pool of worker
threads (zlib releases the GIL) while the next batch is being sampled.

* :func:`save_single_image` -- drop-in for ``runners.utils.save_single_image`` (same signature, identical file).
* :func:`get_image_grid`    -- drop-in for ``runners.utils.get_image_grid`` (``runners/utils.py:77-84``): the grid image the
  runner logs after every validation step (``BBDMRunner.py:205-222``), same bytes, without torchvision.
* :func:`batch_to_uint8`    -- [N, C, H, W] fp32 on the GPU -> [N, H, W, C] uint8 host tensor (pinned).
* :class:`ImageWriter`      -- ``submit(batch, directory, names, to_normal)`` returns at once; ``close()`` joins.
No CPU fallback: the tensor must live on the GPU (the files themselves are written by PIL, as in the reference).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import torch

from . import _lib

__all__ = ["batch_to_uint8", "save_single_image", "save_batch", "get_image_grid", "ImageWriter"]


def _to_u8_device(batch: torch.Tensor, to_normal: bool) -> torch.Tensor:
    _lib.require_gpu(batch)
    if batch.dim() != 4:
        raise ValueError(f"expected [N, C, H, W], got {tuple(batch.shape)}")
    x = batch.detach()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, C, dtype=torch.uint8, device=x.device)
    with _lib.device_guard(x.device):
        _lib.call("bbdm_images_to_u8_f32", x.data_ptr(), out.data_ptr(), N, C, H, W, 1 if to_normal else 0,
                  _lib.current_stream(x.device))
    return out


def batch_to_uint8(batch: torch.Tensor, to_normal: bool = True, non_blocking: bool = False):
    """[N, C, H, W] fp32 device tensor -> uint8 [N, H, W, C] on the host (one kernel + one copy).

    With ``non_blocking`` the copy is only enqueued: returns (host tensor, event) and the caller waits on the event."""
    dev_u8 = _to_u8_device(batch, to_normal)
    if not dev_u8.is_cuda:                       # (the CPU-emulated test back end: already host memory)
        return (dev_u8, None) if non_blocking else dev_u8
    host = torch.empty(dev_u8.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(dev_u8, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev_u8.device))
    if non_blocking:
        host._bbdm_keepalive = dev_u8            # the device buffer must outlive the asynchronous copy
        return host, ev
    ev.synchronize()
    return host


def _write_png(pixels, path: str):
    from PIL import Image
    a = pixels.numpy()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(path)


@torch.no_grad()
def save_single_image(image, save_path, file_name, to_normal=True):
    """runners/utils.py:67-74, same signature: ``image`` is [C, H, W] on the GPU."""
    host = batch_to_uint8(image.unsqueeze(0), to_normal)
    _write_png(host[0], os.path.join(save_path, file_name))


@torch.no_grad()
def save_batch(batch, save_path, file_names: Sequence[str], to_normal=True):
    """All images of ``batch`` ([N, C, H, W]) to ``save_path/file_names[i]`` (synchronous)."""
    host = batch_to_uint8(batch, to_normal)
    for i, name in enumerate(file_names):
        _write_png(host[i], os.path.join(save_path, name))


@torch.no_grad()
def get_image_grid(batch, grid_size=4, to_normal=True):
    """runners/utils.py:77-84, same signature and result: ``torchvision.utils.make_grid(batch, nrow=grid_size)`` (2-pixel zero
    padding around every image, single-channel batches repeated to 3 channels, a single image returned as is) followed by the
    [-1, 1] -> uint8 conversion, as a numpy array [H', W', C].  The conversion runs once over the batch on the GPU (the same fp32
    arithmetic step by step); the grid is assembled from the bytes on the host -- a padding pixel, value 0 before the conversion,
    becomes 0 * 0.5 + 0.5 -> 128 (``to_normal``) or 0."""
    import numpy as np
    if batch.dim() != 4:
        raise ValueError(f"expected [N, C, H, W], got {tuple(batch.shape)}")
    if batch.shape[1] == 1:
        batch = batch.expand(-1, 3, -1, -1)
    px = batch_to_uint8(batch, to_normal).numpy()                    # [N, H, W, C]
    n, h, w, c = px.shape
    if n == 1:
        return px[0].copy()
    pad = 2
    xmaps = min(int(grid_size), n)
    ymaps = -(-n // xmaps)
    grid = np.full((ymaps * (h + pad) + pad, xmaps * (w + pad) + pad, c), 128 if to_normal else 0, dtype=np.uint8)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[y * (h + pad) + pad: y * (h + pad) + pad + h, x * (w + pad) + pad: x * (w + pad) + pad + w] = px[k]
    return grid


class ImageWriter:
    """Asynchronous writer: ``submit`` enqueues conversion + D2H on the current stream and hands the PNG encoding to worker
    threads; the sampling loop continues at once.  ``close()`` (or leaving the ``with`` block) waits for every file and
    re-raises the first worker error."""

    def __init__(self, workers: int = 8, max_pending: int = 64):
        self.pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="bbdm-egress")
        self.pending: List = []
        self.max_pending = max_pending

    def submit(self, batch: torch.Tensor, save_path: str, file_names: Sequence[str], to_normal: bool = True):
        if len(file_names) != batch.shape[0]:
            raise ValueError("one file name per image")
        host, ev = batch_to_uint8(batch, to_normal, non_blocking=True)

        def work(i, name):
            if ev is not None:
                ev.synchronize()
            _write_png(host[i], os.path.join(save_path, name))

        for i, name in enumerate(file_names):
            self.pending.append(self.pool.submit(work, i, name))
        if len(self.pending) > self.max_pending:          # bounded backlog: pinned buffers are held until written
            self._drain(len(self.pending) - self.max_pending)

    def _drain(self, n: Optional[int] = None):
        todo, self.pending = (self.pending, []) if n is None else (self.pending[:n], self.pending[n:])
        for f in todo:
            f.result()

    def flush(self):
        self._drain()

    def close(self):
        try:
            self._drain()
        finally:
            self.pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
