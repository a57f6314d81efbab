"""The in-package PyTorch first stage (bbdm_amd/first_stage.py: the CPU / fallback branch of LatentBrownianBridgeModel.encode / decode)
against the golden fixture the REAL reference VQModel produced (oracle/make_golden_vq.py) -- runs anywhere, no reference checkout
needed.  The HIP first stage is held to the same fixture in tests/test_first_stage_gpu.py."""
import torch

import first_stage_cases as C


def test_pytorch_first_stage_reproduces_the_reference_golden():
    C.golden_vq_f4(torch.device("cpu"), hip=False, tol=2e-5)
