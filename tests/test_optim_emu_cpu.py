"""bbdm_amd.optim (fused Adam + EMA) on the CPU-emulated kernels: parity with torch.optim.Adam and with
the reference's own EMA class (imported from /root/reference when mounted, else an inline restatement of its 8 lines)."""
import pytest
import torch

import optim_cases as C
from emu_backend import emulated_backend

CPU = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emulated_backend() as emu:
        yield emu


@pytest.mark.parametrize("wd,beta1", [(0.0, 0.9), (0.01, 0.5)])
def test_fused_adam_matches_torch_adam(wd, beta1):
    C.adam_parity(CPU, wd, beta1)


def test_ema_matches_reference_ema_and_fuses_into_the_step():
    C.ema_parity(CPU)
