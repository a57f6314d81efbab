"""Sample egress on the CPU-emulated kernels: bytes identical to runners/utils.py:save_single_image."""
import pytest
import torch

import egress_cases as C
from emu_backend import emulated_backend

CPU = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emulated_backend() as emu:
        yield emu


@pytest.mark.parametrize("to_normal", [True, False])
def test_uint8_pixels_are_bit_exact(to_normal):
    C.u8_bit_exact(CPU, to_normal)


def test_png_files_are_byte_identical(tmp_path):
    C.files_byte_identical(CPU, tmp_path)


def test_image_grid_is_identical_to_the_reference():
    C.image_grid_identical(CPU)
