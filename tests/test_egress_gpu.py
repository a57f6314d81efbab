"""Sample egress on the GPU: uint8 pixels bit-exact and PNG files byte-identical to the reference's
save_single_image (runners/utils.py:67-74); whole-batch conversion rate printed."""
import pytest
import torch

import egress_cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("to_normal", [True, False])
def test_uint8_pixels_are_bit_exact(dev, to_normal):
    C.u8_bit_exact(dev, to_normal)


def test_png_files_are_byte_identical(dev, tmp_path):
    C.files_byte_identical(dev, tmp_path)


def test_image_grid_is_identical_to_the_reference(dev):
    """get_image_grid (runners/utils.py:77-84, used at BBDMRunner.py:205-222): same array as make_grid + the uint8 conversion."""
    C.image_grid_identical(dev)


def test_batch16_256x256_rate(dev):
    from bbdm_amd import egress
    x = torch.randn(16, 3, 256, 256, device=dev).clamp(-1, 1)
    egress.batch_to_uint8(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        h = egress.batch_to_uint8(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    import time
    t0 = time.perf_counter()
    for i in range(16):                                  # the reference's per-image path on the same batch
        C.reference_u8(x[i])
    ref_ms = (time.perf_counter() - t0) * 1e3
    print(f"egress: 16 x 3 x 256 x 256 -> uint8 host in {ms:.3f} ms (one kernel + one D2H); reference per-image path "
          f"{ref_ms:.3f} ms")
    assert torch.equal(h[3], C.reference_u8(x[3].cpu()))
