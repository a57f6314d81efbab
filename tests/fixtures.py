"""Helpers shared by the tests: load a golden fixture and rebuild its weights (test infrastructure)."""
import contextlib
import os

import torch

from fixture_weights import synth_weights          # tests/fixture_weights.py
import bbdm_oracle as O                            # oracle/

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("tiny_concat", "tiny_nocond", "tiny_ysubx")
# + SpatialTransformer / cross-attention conditioning: sampling, and -- by name -- the training tests
# (gradients, accumulation, the two-rank DDP steps: its 3-channel context gives channel-padded to_k / to_v weight gradients)
INFER_CASES = CASES + ("tiny_xattn",)


def load_case(name):
    rec = torch.load(os.path.join(GOLDEN, f"{name}.pt"), weights_only=False)
    unet_sd = synth_weights(rec["unet_shapes"], rec["weight_seed"])
    sd = dict(rec["buffers"])                       # reference order: schedule buffers first, then the UNet
    sd.update({"denoise_fn." + k: v for k, v in unet_sd.items()})
    rec["state_dict"] = sd
    return rec


def oracle_model(rec):
    spec = O.UNetSpec(**rec["unet_params"])
    bb = rec["bb_params"]
    return O.OracleBBDM(rec["state_dict"], spec, num_timesteps=bb["num_timesteps"], mt_type=bb["mt_type"],
                        max_var=bb.get("max_var", 1), eta=bb.get("eta", 1), skip_sample=bb["skip_sample"],
                        sample_type=bb["sample_type"], sample_step=bb["sample_step"], loss_type=bb["loss_type"],
                        objective=bb["objective"])


@contextlib.contextmanager
def few_threads(n=32):
    """Run the CPU oracle on at most ``n`` threads: its small problems (one 64x64 image, tiny golden models) are SLOWER on the 128
    default threads of the GPU box than on a few (oversubscription), and they dominated the GPU suite's wall time."""
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(old, n)))
    try:
        yield
    finally:
        torch.set_num_threads(old)


def rel_err(a, b):
    """max|a-b| / max|b|  (SURVEY.md §8c per-step parity metric, max-norm form)."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2  (SURVEY.md §8c per-step parity metric, L2 form: an error spread over many small-magnitude
    elements passes the max-norm form unnoticed)."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def parity_err(a, b):
    """BOTH metrics of SURVEY.md §8c as one number, the larger of max|a-b| / max|b| and ||a-b||_2 / ||b||_2:
    ``parity_err(a, b) < tol`` asserts each of them against the bar."""
    return max(rel_err(a, b), rel_l2(a, b))
