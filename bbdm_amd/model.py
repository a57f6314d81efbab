"""BrownianBridgeModel / LatentBrownianBridgeModel -- drop-in mirrors of the reference classes, MI355X-native.

Reference: model/BrownianBridge/BrownianBridgeModel.py:15-225 and LatentBrownianBridgeModel.py:19-147.
The runner (runners/DiffusionBasedModelRunners/BBDMRunner.py:21-29, 164-253) touches the model only through
``Model(config.model)``, ``.apply(weights_init)``, ``.get_parameters()``, ``net(x, x_cond) -> (loss, dict)``,
``.sample(...)``, ``.encode(...)``, ``state_dict()/load_state_dict()`` and ``named_parameters()`` -- all kept with
the same names, argument meaning and error behaviour (asserts / NotImplementedError at the same places).

What changed underneath: ``denoise_fn`` is :class:`bbdm_amd.unet.UNetModel` (HIP kernels), and the scheduler
arithmetic (q_sample / predict_x0 / the p_sample update / the loss) are fused HIP kernels reached through the
C-ABI of ``include/bbdm_hip.h``.  Random numbers still come from torch's generator (``torch.randn_like`` /
``torch.randint``), so ``main.py``'s seeding (main.py:57-65) keeps its meaning.  No CPU fallback exists.
"""
from __future__ import annotations

import itertools
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .unet import UNetModel

try:                                   # tqdm is what the reference wraps its loops in; optional here
    from tqdm.autonotebook import tqdm
except Exception:                      # pragma: no cover
    def tqdm(it, **kw):
        return it

_OBJECTIVES = {"grad": 0, "noise": 1, "ysubx": 2}
_LOSSES = {"l1": 0, "l2": 1}


def _need_gpu(*ts):
    _lib.require_gpu(*ts)


def _launch(t: torch.Tensor, name: str, *args):
    """One C-ABI call on ``t``'s device and torch's current stream there."""
    with _lib.device_guard(t.device):
        _lib.call(name, *args, _lib.current_stream(t.device))


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def bridge_schedule(T, mt_type, max_var, skip_sample, sample_type, sample_step):
    """The six [T] schedule tables (float64 numpy, in registration order) and the sampling step table.

    m_t: 'linear' = linspace(0.001, 0.999); 'sin' = 1.0075**linspace(0, T) normalised with the last entry pinned to
    0.999.  variance_t = 2 (m_t - m_t^2) max_var.  The "*_tminus" tables are the same shifted by one step with a
    leading 0.  (BrownianBridgeModel.py:45-59; only m_t and variance_t are read by the sampler.)
    Step table (BrownianBridgeModel.py:69-79): skip+'linear' = arange(T-1, 1, -(T-1)/(sample_step-2)).long() ++ [1, 0];
    skip+'cosine' as the reference defines it (float64 values); no skip = T-1 .. 0.
    """
    if mt_type == "linear":
        m = np.linspace(0.001, 0.999, T)
    elif mt_type == "sin":
        m = np.power(1.0075, np.linspace(0, T, T))
        m /= m[-1]
        m[-1] = 0.999
    else:
        raise NotImplementedError
    shift = lambda a, first: np.concatenate(([first], a[:-1]))
    m_prev = shift(m, 0)
    var = 2. * (m - m ** 2) * max_var
    var_prev = shift(var, 0.)
    var_t_prev = var - var_prev * ((1. - m) / (1. - m_prev)) ** 2
    tables = {
        "m_t": m, "m_tminus": m_prev, "variance_t": var, "variance_tminus": var_prev,
        "variance_t_tminus": var_t_prev, "posterior_variance_t": var_t_prev * var_prev / var,
    }
    steps = None
    if not skip_sample:
        steps = torch.arange(T - 1, -1, -1)
    elif sample_type == "linear":
        stride = (T - 1) / (sample_step - 2)
        steps = torch.cat((torch.arange(T - 1, 1, step=-stride).long(), torch.tensor([1, 0], dtype=torch.long)))
    elif sample_type == "cosine":
        grid = np.linspace(start=0, stop=T, num=sample_step + 1)
        steps = torch.from_numpy((np.cos(grid / T * np.pi) + 1.) / 2. * T)
    return tables, steps


class BrownianBridgeModel(nn.Module):
    """BrownianBridgeModel.py:15-225."""

    def __init__(self, model_config):
        super().__init__()
        self.model_config = model_config
        model_params = model_config.BB.params
        self.num_timesteps = model_params.num_timesteps
        self.mt_type = model_params.mt_type
        self.max_var = model_params.max_var if model_params.__contains__("max_var") else 1
        self.eta = model_params.eta if model_params.__contains__("eta") else 1
        self.skip_sample = model_params.skip_sample
        self.sample_type = model_params.sample_type
        self.sample_step = model_params.sample_step
        self.steps = None
        self.register_schedule()

        self.loss_type = model_params.loss_type
        self.objective = model_params.objective

        self.image_size = model_params.UNetParams.image_size
        self.channels = model_params.UNetParams.in_channels
        self.condition_key = model_params.UNetParams.condition_key

        self.denoise_fn = UNetModel(**vars(model_params.UNetParams))

    # ------------------------------------------------------------------------------------------------------
    def register_schedule(self):
        """Schedule buffers + sampling step table (BrownianBridgeModel.py:42-79).  Host-side float64 -> fp32 buffers
        that are part of the ``state_dict``; ``self.steps`` stays a CPU tensor attribute as in the reference."""
        tables, self.steps = bridge_schedule(self.num_timesteps, self.mt_type, self.max_var, self.skip_sample,
                                             self.sample_type, self.sample_step)
        for name, values in tables.items():
            self.register_buffer(name, torch.tensor(values, dtype=torch.float32))
        self._steps_list = None

    def _steps_host(self):
        """``steps`` as a Python list (it is a CPU tensor attribute in the reference too: no device sync)."""
        if self._steps_list is None or len(self._steps_list) != len(self.steps):
            self._steps_list = [int(s) for s in self.steps]
        return self._steps_list

    def apply(self, weight_init):
        self.denoise_fn.apply(weight_init)
        return self

    def get_parameters(self):
        return self.denoise_fn.parameters()

    # ------------------------------------------------------------------------------------------------------
    def forward(self, x, y, context=None):
        """BrownianBridgeModel.py:88-96."""
        if self.condition_key == "nocond":
            context = None
        else:
            context = y if context is None else context
        b, c, h, w, device, img_size, = *x.shape, x.device, self.image_size
        assert h == img_size and w == img_size, f'height and width of image must be {img_size}'
        t = torch.randint(0, self.num_timesteps, (b,), device=device).long()
        return self.p_losses(x, y, context, t)

    def p_losses(self, x0, y, context, t, noise=None):
        """BrownianBridgeModel.py:98-126."""
        if self.loss_type not in _LOSSES:
            raise NotImplementedError()
        noise = torch.randn_like(x0) if noise is None else noise
        x_t, objective = self.q_sample(x0, y, t, noise)
        objective_recon = self.denoise_fn(x_t, timesteps=t, context=context)
        if objective_recon.requires_grad:
            from .autograd import bb_loss      # differentiable HIP loss (training path)
            recloss = bb_loss(objective, objective_recon, self.loss_type)
        else:
            recloss = self._loss(objective, objective_recon)
        x0_recon = self.predict_x0_from_objective(x_t, y, t, objective_recon.detach())
        log_dict = {"loss": recloss, "x0_recon": x0_recon}
        return recloss, log_dict

    def _loss(self, a, b):
        _need_gpu(a, b)
        a, b = _f32c(a), _f32c(b)
        partial_ = torch.zeros(4, dtype=torch.float64, device=a.device)       # one exact limb cell (csrc/stats_acc.h)
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        _launch(a, "bbdm_bb_loss_f32", a.data_ptr(), b.data_ptr(), partial_.data_ptr(), out.data_ptr(), a.numel(),
                  _LOSSES[self.loss_type])
        return out[0]

    def q_sample(self, x0, y, t, noise=None):
        """BrownianBridgeModel.py:128-146 -> (x_t, objective)."""
        if self.objective not in _OBJECTIVES:
            raise NotImplementedError()
        noise = torch.randn_like(x0) if noise is None else noise
        _need_gpu(x0, y, noise, t)
        x0c, yc, nc = _f32c(x0), _f32c(y), _f32c(noise)
        tc = t.to(torch.int64).contiguous()
        x_t, target = torch.empty_like(x0c), torch.empty_like(x0c)
        _launch(x0c, "bbdm_bb_q_sample_f32", x0c.data_ptr(), yc.data_ptr(), nc.data_ptr(), tc.data_ptr(),
                  self.m_t.data_ptr(), self.variance_t.data_ptr(), x_t.data_ptr(), target.data_ptr(),
                  x0c.shape[0], x0c[0].numel(), _OBJECTIVES[self.objective])
        return x_t, target

    def predict_x0_from_objective(self, x_t, y, t, objective_recon):
        """BrownianBridgeModel.py:148-160."""
        if self.objective not in _OBJECTIVES:
            raise NotImplementedError
        _need_gpu(x_t, y, objective_recon, t)
        a, b, p = _f32c(x_t), _f32c(y), _f32c(objective_recon)
        tc = t.to(torch.int64).contiguous()
        out = torch.empty_like(a)
        _launch(a, "bbdm_bb_predict_x0_f32", a.data_ptr(), b.data_ptr(), p.data_ptr(), tc.data_ptr(),
                  self.m_t.data_ptr(), self.variance_t.data_ptr(), out.data_ptr(), a.shape[0], a[0].numel(),
                  _OBJECTIVES[self.objective])
        return out

    @torch.no_grad()
    def q_sample_loop(self, x0, y):
        """BrownianBridgeModel.py:162-169."""
        imgs = [x0]
        for i in tqdm(range(self.num_timesteps), desc='q sampling loop', total=self.num_timesteps):
            t = torch.full((y.shape[0],), i, device=x0.device, dtype=torch.long)
            img, _ = self.q_sample(x0, y, t)
            imgs.append(img)
        return imgs

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def p_sample(self, x_t, y, context, i, clip_denoised=False):
        """BrownianBridgeModel.py:171-201: one UNet call + ONE fused update kernel -> (x_{t-1}, x0_recon)."""
        if self.objective not in _OBJECTIVES:
            raise NotImplementedError
        _need_gpu(x_t, y)
        steps = self._steps_host()
        step = steps[i]
        nxt = 0 if step == 0 else steps[i + 1]
        if not (0 <= step < self.num_timesteps and 0 <= nxt < self.num_timesteps):
            # the reference gathers m_t[t] and fails there (sample_type 'cosine' starts at t = num_timesteps:
            # BrownianBridgeModel.py:74-77); the kernels read the tables unchecked, so the check lives here
            raise IndexError(f"timestep {max(step, nxt)} is out of range for the {self.num_timesteps}-entry schedule")
        x_t, y = _f32c(x_t), _f32c(y)
        fn = self.denoise_fn
        plan = None
        if isinstance(fn, UNetModel):
            # the prediction is consumed by the fused kernel below at once (no per-step copy of it); the timestep is a host integer
            # (one fill of the plan's buffer instead of torch.full + copy); the plan skips the copies of inputs it already holds
            objective_recon, plan = fn.infer_step(x_t, step, context)
        else:
            t = torch.full((x_t.shape[0],), step, device=x_t.device, dtype=torch.long)
            objective_recon = fn(x_t, timesteps=t, context=context)
        is_last = step == 0
        noise = None if is_last else torch.randn_like(x_t)
        x_next, x0_recon = torch.empty_like(x_t), torch.empty_like(x_t)
        # the loop feeds x_next back as the next x_t (BrownianBridgeModel.py:218-220): the step writes it into the plan's input buffer too
        alias = None if (plan is None or is_last) else plan.x_in
        _launch(x_t, "bbdm_bb_p_sample_step_f32", x_t.data_ptr(), y.data_ptr(), objective_recon.data_ptr(),
                  None if noise is None else noise.data_ptr(), self.m_t.data_ptr(), self.variance_t.data_ptr(),
                  step, 0 if is_last else steps[i + 1], 1 if is_last else 0, float(self.eta),
                  1 if clip_denoised else 0, _OBJECTIVES[self.objective], x_next.data_ptr(), x0_recon.data_ptr(),
                  None if alias is None else alias.data_ptr(), x_t.shape[0], x_t[0].numel())
        if alias is not None:
            plan.holds_input(x_next)
        if is_last:
            return x0_recon, x0_recon
        return x_next, x0_recon

    @torch.no_grad()
    def p_sample_loop(self, y, context=None, clip_denoised=True, sample_mid_step=False):
        """BrownianBridgeModel.py:203-221."""
        if self.condition_key == "nocond":
            context = None
        else:
            context = y if context is None else context

        if sample_mid_step:
            imgs, one_step_imgs = [y], []
            for i in tqdm(range(len(self.steps)), desc=f'sampling loop time step', total=len(self.steps)):
                img, x0_recon = self.p_sample(x_t=imgs[-1], y=y, context=context, i=i, clip_denoised=clip_denoised)
                imgs.append(img)
                one_step_imgs.append(x0_recon)
            return imgs, one_step_imgs
        else:
            img = y
            for i in tqdm(range(len(self.steps)), desc=f'sampling loop time step', total=len(self.steps)):
                img, _ = self.p_sample(x_t=img, y=y, context=context, i=i, clip_denoised=clip_denoised)
            return img

    @torch.no_grad()
    def sample(self, y, context=None, clip_denoised=True, sample_mid_step=False):
        """BrownianBridgeModel.py:223-225."""
        return self.p_sample_loop(y, context, clip_denoised, sample_mid_step)


def disabled_train(self, mode=True):
    """Bound over ``vqgan.train`` so later ``.train()`` calls cannot un-freeze it (LatentBrownianBridgeModel.py:13-16)."""
    return self


class LatentBrownianBridgeModel(BrownianBridgeModel):
    """LatentBrownianBridgeModel.py:19-147: the same bridge run in the latent space of a frozen VQGAN.

    Only the wrapper API lives here (constructor, ``forward``, ``encode``/``decode``, ``sample``, the externally
    assigned ``ori_latent_mean/std`` and ``cond_latent_mean/std`` attributes -- BBDMRunner.py:41-44).  The first stage
    is any module exposing ``encoder / quant_conv / quantize / decode``: by default ``bbdm_amd.first_stage_hip.VQModel``
    built from ``model_config.VQGAN.params`` (encode / decode on the HIP kernels for GPU tensors; loads the reference's VQGAN
    checkpoints), or whatever is passed as ``vqgan=`` (e.g. the checkout's own ``model.VQGAN.vqgan.VQModel``).  The first stage
    is the ONE place with a PyTorch branch: a foreign ``vqgan`` without ``encode_latent`` / ``decode_latent``, or CPU tensors, go
    through the module's own PyTorch ``encoder`` / ``quantize`` / ``decode`` (north_star leaves the VQGAN on PyTorch-ROCm); the UNet
    and the bridge have none.
    """

    def __init__(self, model_config, vqgan: nn.Module = None):
        super().__init__(model_config)
        if vqgan is None:
            # state_dict-compatible with the reference's VQModel; encode / decode run on the HIP kernels
            from .first_stage_hip import VQModel
            vqgan = VQModel(**vars(model_config.VQGAN.params))
        self.vqgan = vqgan.eval()
        self.vqgan.train = disabled_train
        self.vqgan.requires_grad_(False)
        print(f"load vqgan from {getattr(model_config.VQGAN.params, 'ckpt_path', None)}")

        key = self.condition_key
        if key == 'nocond':
            self.cond_stage_model = None
        elif key == 'first_stage':
            self.cond_stage_model = self.vqgan
        elif key == 'SpatialRescaler':
            from .cond_stage import SpatialRescaler          # same constructor / state_dict as the reference's class
            self.cond_stage_model = SpatialRescaler(**vars(model_config.CondStageParams))
        else:
            raise NotImplementedError

    def get_ema_net(self):
        return self

    def get_parameters(self):
        unet = self.denoise_fn.parameters()
        if self.condition_key != 'SpatialRescaler':
            print("get parameters to optimize: UNet")
            return unet
        print("get parameters to optimize: SpatialRescaler, UNet")
        return itertools.chain(unet, self.cond_stage_model.parameters())

    def apply(self, weights_init):
        BrownianBridgeModel.apply(self, weights_init)
        if self.cond_stage_model is not None:
            self.cond_stage_model.apply(weights_init)
        return self

    def forward(self, x, x_cond, context=None):
        # both images go through the frozen encoder without a graph; only the UNet (+ rescaler) is trained
        with torch.no_grad():
            latents = [self.encode(img, cond=flag).detach() for img, flag in ((x, False), (x_cond, True))]
        return BrownianBridgeModel.forward(self, latents[0], latents[1], self.get_cond_stage_context(x_cond))

    def get_cond_stage_context(self, x_cond):
        if self.cond_stage_model is None:
            return None
        context = self.cond_stage_model(x_cond)
        return context.detach() if self.condition_key == 'first_stage' else context

    def _latent_stats(self, cond):
        if cond:
            return self.cond_latent_mean, self.cond_latent_std
        return self.ori_latent_mean, self.ori_latent_std

    def _use_norm(self, normalize):
        return self.model_config.normalize_latent if normalize is None else normalize

    @torch.no_grad()
    def encode(self, x, cond=True, normalize=None):
        if hasattr(self.vqgan, "encode_latent") and x.is_cuda:      # bbdm_amd.first_stage_hip: the whole encoder on HIP
            z = self.vqgan.encode_latent(x, quant_conv=not self.model_config.latent_before_quant_conv)
        else:
            z = self.vqgan.encoder(x)
            if not self.model_config.latent_before_quant_conv:
                z = self.vqgan.quant_conv(z)
        if self._use_norm(normalize):
            mean, std = self._latent_stats(cond)
            z = (z - mean) / std
        return z

    @torch.no_grad()
    def decode(self, x_latent, cond=True, normalize=None):
        z = x_latent
        if self._use_norm(normalize):
            mean, std = self._latent_stats(cond)
            z = z * std + mean
        if hasattr(self.vqgan, "decode_latent") and z.is_cuda:      # quantize + post_quant_conv + decoder on HIP
            return self.vqgan.decode_latent(z, quant_conv_first=self.model_config.latent_before_quant_conv)
        if self.model_config.latent_before_quant_conv:
            z = self.vqgan.quant_conv(z)
        z_q, _, _ = self.vqgan.quantize(z)
        return self.vqgan.decode(z_q)

    @torch.no_grad()
    def sample(self, x_cond, clip_denoised=False, sample_mid_step=False):
        result = self.p_sample_loop(y=self.encode(x_cond, cond=True), context=self.get_cond_stage_context(x_cond),
                                    clip_denoised=clip_denoised, sample_mid_step=sample_mid_step)
        if not sample_mid_step:
            return self.decode(result, cond=False)

        def decode_all(latents, desc):
            return [self.decode(z.detach(), cond=False).to('cpu')
                    for z in tqdm(latents, initial=0, desc=desc, dynamic_ncols=True, smoothing=0.01)]

        trajectory, one_step = result
        return (decode_all(trajectory, "save output sample mid steps"),
                decode_all(one_step, "save one step sample mid steps"))

    @torch.no_grad()
    def sample_vqgan(self, x):
        return self.vqgan(x)[0]
