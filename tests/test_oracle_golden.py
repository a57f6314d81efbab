"""The oracle (oracle/bbdm_oracle.py) against the golden vectors produced by the real reference."""
import os

import pytest
import torch

import bbdm_oracle as O
from fixtures import INFER_CASES as CASES, GOLDEN, load_case, oracle_model, rel_err

TOL = 2e-5      # oracle and reference run the same ATen ops on CPU; only op grouping differs


def test_schedule_known_answers():
    kat = torch.load(os.path.join(GOLDEN, "schedule_kat.pt"), weights_only=False)
    for (mt, skip), ref in kat.items():
        bufs, steps = O.make_schedule(1000, mt, 1.0, skip, "linear", 200)
        assert torch.equal(steps, ref["steps"])
        for k, v in bufs.items():
            assert torch.equal(v, ref[k]) or torch.allclose(v, ref[k], rtol=0, atol=0, equal_nan=True), (mt, skip, k)
    # SURVEY.md §8c frozen facts
    b, steps = O.make_schedule(1000, "linear", 1.0, True, "linear", 200)
    assert len(steps) == 200 and int(steps.sum()) == 99307 and steps[:4].tolist() == [999, 993, 988, 983]
    assert steps[-4:].tolist() == [10, 5, 1, 0]
    assert abs(float(b["m_t"][0]) - 0.0010000000475) < 1e-12 and abs(float(b["m_t"][-1]) - 0.9990000129) < 1e-9
    assert abs(float(b["variance_t"].max()) - 0.4999994934) < 1e-9


@pytest.mark.parametrize("name", CASES)
def test_unet_matches_reference_golden(name):
    rec = load_case(name)
    m = oracle_model(rec)
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else rec["y"]
    with torch.no_grad():
        out = m.denoise(rec["x0"], rec["t"], ctx)
    assert rel_err(out, rec["unet_out"]) < TOL


@pytest.mark.parametrize("name", CASES)
def test_q_sample_and_loss(name):
    rec = load_case(name)
    m = oracle_model(rec)
    x_t, obj = O.q_sample(m.bufs, rec["x0"], rec["y"], rec["t"], rec["noise"], m.objective)
    assert torch.equal(x_t, rec["q_x_t"]) and torch.equal(obj, rec["q_objective"])
    with torch.no_grad():
        loss, log = m.p_losses(rec["x0"], rec["y"], None, rec["t"], rec["noise"])
    assert abs(float(loss) - float(rec["loss"])) < TOL * max(1.0, abs(float(rec["loss"])))
    assert rel_err(log["x0_recon"], rec["x0_recon"]) < TOL


@pytest.mark.parametrize("name", CASES)
def test_p_sample_steps(name):
    rec = load_case(name)
    m = oracle_model(rec)
    ctx = None if rec["unet_params"]["condition_key"] == "nocond" else rec["y"]
    for clip, i, a_ref, b_ref in rec["p_out"]:
        a, b = m.p_sample(rec["p_x_t"], rec["y"], ctx, i, clip_denoised=clip, noise=rec["p_eps"])
        assert rel_err(a, a_ref) < TOL and rel_err(b, b_ref) < TOL, (clip, i)


@pytest.mark.parametrize("name", CASES)
def test_free_running_loop(name):
    rec = load_case(name)
    m = oracle_model(rec)
    n = len(m.steps)
    out = m.p_sample_loop(rec["y"], None, clip_denoised=True, noises=[rec["p_eps"]] * n)
    assert rel_err(out, rec["loop_out"]) < 1e-3      # free-running drift over up to 50 steps
