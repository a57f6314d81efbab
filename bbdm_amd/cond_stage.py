"""Conditioning-stage encoders of ``LatentBrownianBridgeModel`` (``condition_key`` -> ``cond_stage_model``).

``SpatialRescaler`` (model/BrownianBridge/base/modules/encoders/modules.py:106-134 of the reference): the conditioning image
resized ``n_stages`` times by ``multiplier`` and optionally mapped to ``out_channels`` by a (trainable) 1x1 convolution.  It
runs ONCE per batch on a 3-8 channel image -- not on the per-step hot path -- so it is plain PyTorch-ROCm: what matters is
that a config with ``condition_key: SpatialRescaler`` needs nothing from the reference checkout, that checkpoints load
(``channel_mapper.weight`` / ``.bias``) and that its parameters train (``get_parameters``, LatentBrownianBridgeModel.py:43-50;
their gradient arrives through the UNet's d context: the concat slice + the cross-attention keys / values).
"""
from __future__ import annotations

import torch.nn as nn
import torch.nn.functional as F

__all__ = ["SpatialRescaler"]

_METHODS = ("nearest", "linear", "bilinear", "trilinear", "bicubic", "area")


class SpatialRescaler(nn.Module):
    def __init__(self, n_stages=1, method="bilinear", multiplier=0.5, in_channels=3, out_channels=None, bias=False):
        super().__init__()
        if n_stages < 0:
            raise AssertionError("n_stages must be >= 0")
        if method not in _METHODS:
            raise AssertionError(f"method must be one of {_METHODS}")
        self.n_stages, self.method, self.multiplier = n_stages, method, multiplier
        self.remap_output = out_channels is not None
        if self.remap_output:
            print(f"Spatial Rescaler mapping from {in_channels} to {out_channels} channels after resizing.")
            self.channel_mapper = nn.Conv2d(in_channels, out_channels, 1, bias=bias)

    def forward(self, x):
        for _ in range(self.n_stages):
            x = F.interpolate(x, scale_factor=self.multiplier, mode=self.method)
        return self.channel_mapper(x) if self.remap_output else x

    def encode(self, x):
        return self(x)
