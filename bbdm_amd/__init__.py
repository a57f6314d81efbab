"""bbdm_amd -- the MI355X (gfx950) hot path of xuekt98/BBDM: denoising UNet + Brownian-Bridge scheduler as
hand-written HIP kernels behind a C-ABI (include/bbdm_hip.h), exposed through drop-in model classes.

    from bbdm_amd import BrownianBridgeModel, LatentBrownianBridgeModel      # same API as the reference classes
"""
from .model import BrownianBridgeModel, LatentBrownianBridgeModel, bridge_schedule  # noqa: F401
from .unet import UNetModel  # noqa: F401
from .cond_stage import SpatialRescaler  # noqa: F401

__all__ = ["BrownianBridgeModel", "LatentBrownianBridgeModel", "UNetModel", "SpatialRescaler", "bridge_schedule"]
