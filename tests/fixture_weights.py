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
