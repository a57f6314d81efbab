"""Deterministic synthetic UNet weights for fixtures and benchmarks  --  TEST INFRASTRUCTURE.

``numpy.random.RandomState`` streams are frozen by numpy's compatibility policy, so the same
(key, shape) list and seed give bit-identical weights in the build container (where the golden
outputs are produced from the real reference) and on the GPU box (where they are consumed).

Values follow the reference's ``weights_init`` (runners/utils.py:35-45: N(0, 0.02) on Conv2d/Linear
weights) but ALSO randomise what ``weights_init`` leaves at zero/one (Conv1d qkv / proj_out, the
zero_module convs, biases, GroupNorm affine) so that no branch of the graph is identically zero
(a zero ``proj_out`` would test the attention kernel vacuously).
"""
from __future__ import annotations

import numpy as np
import torch


def synth_weights(shapes, seed: int, w_std: float = 0.04):
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in shapes:
        n = int(np.prod(shape)) if len(shape) else 1
        v = rs.standard_normal(n).astype(np.float32).reshape(shape)
        if len(shape) > 1:                                   # conv / linear / conv1d weights
            v *= (2.5 * w_std if ("proj_out" in key or "qkv" in key) else w_std)
        elif key.endswith("weight"):                         # GroupNorm gamma
            v = 1.0 + 0.1 * v
        else:                                                # biases / GroupNorm beta
            v *= 0.05
        out[key] = torch.from_numpy(np.ascontiguousarray(v))
    return out


STRESS_KINDS = ("dc", "gain3", "student", "smooth")


def stress_weights(shapes, seed: int, kind: str, w_std: float = 0.02):
    """Non-Gaussian weight sets for the parity stress tests (round-5 verdict: the rounding error of the large Winograd tiles depends on
    the data, and a trained checkpoint is not a zero-mean Gaussian).  Starting from ``synth_weights``:
      dc      : every 3x3 filter gets a DC component, w += 0.05 (2.5 sigma): coherent sums over Cin x 9 taps
      gain3   : every conv / linear weight 3x larger (w_std = 0.06)
      student : the 3x3 filters are Student-t (nu = 3) instead of Gaussian: heavy tails, a few taps 10 - 30 sigma out
      smooth  : the 3x3 filters are low-pass (a [1 2 1] x [1 2 1] / 16 kernel times a Gaussian channel-mixing matrix, plus 20 % noise):
                the structured, spatially correlated filters a trained network has
    """
    assert kind in STRESS_KINDS, kind
    out = synth_weights(shapes, seed, w_std=3 * w_std if kind == "gain3" else w_std)
    rs = np.random.RandomState(seed + 1)
    for key, shape in shapes:
        if len(shape) != 4 or shape[2] != 3:
            continue
        w = out[key]
        if kind == "dc":
            out[key] = w + 0.05
        elif kind == "student":
            t = rs.standard_t(3.0, size=shape).astype(np.float32)
            out[key] = torch.from_numpy(np.ascontiguousarray(t * w_std))
        elif kind == "smooth":
            k1 = np.array([1.0, 2.0, 1.0], dtype=np.float32)
            lp = np.outer(k1, k1) / 16.0
            mix = rs.standard_normal(shape[:2]).astype(np.float32) * (4.0 * w_std)
            out[key] = torch.from_numpy(np.ascontiguousarray(mix[:, :, None, None] * lp[None, None])) + 0.2 * w
    return out
