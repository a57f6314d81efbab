"""VQGAN first stage on the CPU-emulated HIP kernels."""
import pytest
import torch

import first_stage_cases as C
from emu_backend import emulated_backend

CPU = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emulated_backend() as emu:
        yield emu


def test_vq_indices_bit_exact():
    C.vq_indices_bit_exact(CPU)


def test_encode_decode_match_the_pytorch_first_stage():
    C.encode_decode_parity(CPU, N=1)
