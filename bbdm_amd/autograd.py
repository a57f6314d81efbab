"""Training path (autograd over the HIP kernels) -- placeholder until the backward kernels land.

The inference path (``torch.no_grad()``: sampling, validation loss) is complete; calling the model with gradients
enabled raises here instead of silently running some other implementation.
"""


def unet_apply(model, x, timesteps, context):
    raise NotImplementedError(
        "bbdm_amd: the UNet backward kernels (conv dgrad/wgrad, GroupNorm/SiLU/FiLM, attention) are not implemented "
        "yet; run under torch.no_grad() (sampling / validation), or freeze the UNet parameters")


def bb_loss(target, pred, loss_type):
    raise NotImplementedError("bbdm_amd: differentiable loss arrives with the backward kernels")
