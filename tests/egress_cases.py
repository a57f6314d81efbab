"""Shared bodies of the sample-egress tests (GPU: test_egress_gpu.py; CPU-emulated kernels: test_egress_emu_cpu.py)."""
import hashlib
import os

import torch


def reference_u8(image, to_normal=True):
    """runners/utils.py:67-74 up to the PIL call, on the CPU (the reference's own sequence of in-place ops)."""
    image = image.detach().clone()
    if to_normal:
        image = image.mul_(0.5).add_(0.5).clamp_(0, 1.)
    return image.mul_(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8)


def reference_save_single_image(image, save_path, file_name, to_normal=True):
    from PIL import Image
    Image.fromarray(reference_u8(image, to_normal).numpy()).save(os.path.join(save_path, file_name))


def make_batch(N=5, C=3, H=20, W=28, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g) * 0.7
    flat = x.view(-1)
    # the rounding boundaries of the 8-bit quantisation: k/255 - 1/510 mapped back to [-1, 1], +- 1 ulp, and the clamps
    k = torch.arange(0, 256, dtype=torch.float32)
    edges = (k - 0.5) / 255.0 * 2.0 - 1.0
    special = torch.cat([edges, torch.nextafter(edges, torch.tensor(2.0)), torch.nextafter(edges, torch.tensor(-2.0)),
                         torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, -0.0, 1e-9, float("inf"), -float("inf")])])
    k = min(special.numel(), flat.numel())
    flat[:k] = special[:k]
    return x


def u8_bit_exact(dev, to_normal):
    from bbdm_amd import egress
    x = make_batch()
    host = egress.batch_to_uint8(x.to(dev), to_normal)
    assert host.dtype == torch.uint8 and tuple(host.shape) == (5, 20, 28, 3)
    for i in range(x.shape[0]):
        assert torch.equal(host[i], reference_u8(x[i], to_normal)), i
    y = make_batch(2, 1, 29, 31, seed=3)                      # single-channel images
    h1 = egress.batch_to_uint8(y.to(dev), to_normal)
    assert torch.equal(h1[1], reference_u8(y[1], to_normal))


def files_byte_identical(dev, tmp_path):
    from bbdm_amd import egress
    x = make_batch(6, 3, 16, 16, seed=5)
    ref_dir, out_dir, out2 = tmp_path / "ref", tmp_path / "one", tmp_path / "async"
    for d in (ref_dir, out_dir, out2):
        d.mkdir()
    names = [f"img_{i}.png" for i in range(6)]
    for i, n in enumerate(names):
        reference_save_single_image(x[i], str(ref_dir), n)
        egress.save_single_image(x[i].to(dev), str(out_dir), n)          # the drop-in, same signature
    with egress.ImageWriter(workers=3, max_pending=4) as w:              # a backlog smaller than the batch: drains midway
        w.submit(x.to(dev), str(out2), names)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    for n in names:
        assert sha(ref_dir / n) == sha(out_dir / n) == sha(out2 / n), n


def make_grid_reference(tensor, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid (the call of runners/utils.py:79; torchvision is not installed here), restated from its source:
    single-channel batches repeated to 3 channels, one image returned as is, otherwise images copied into a pad_value canvas."""
    import math
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * height + padding, height - padding).narrow(2, x * width + padding, width - padding).copy_(tensor[k])
            k += 1
    return grid


def reference_get_image_grid(batch, grid_size=4, to_normal=True):
    """runners/utils.py:77-84 with make_grid restated above."""
    batch = batch.detach().clone()
    image_grid = make_grid_reference(batch, nrow=grid_size)
    if to_normal:
        image_grid = image_grid.mul_(0.5).add_(0.5).clamp_(0, 1.)
    return image_grid.mul_(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()


def image_grid_identical(dev):
    import numpy as np
    from bbdm_amd import egress
    for N, C, H, W, gs, tn in ((5, 3, 20, 28, 4, True), (8, 3, 16, 16, 4, True), (3, 1, 9, 11, 2, True), (1, 3, 8, 8, 4, True),
                               (6, 3, 12, 10, 4, False), (2, 3, 8, 8, 5, True)):
        x = make_batch(N, C, H, W, seed=N + H)
        got = egress.get_image_grid(x.to(dev), grid_size=gs, to_normal=tn)
        ref = reference_get_image_grid(x, grid_size=gs, to_normal=tn)
        assert isinstance(got, np.ndarray) and got.dtype == np.uint8 and got.shape == ref.shape, (got.shape, ref.shape)
        assert np.array_equal(got, ref), (N, C, H, W, gs, tn)
