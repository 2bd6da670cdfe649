"""DDPM / DDIM schedulers for the sampling loop (call sites EMOAnimationPipeline.py:653-654,764,817).

The reference uses `diffusers` schedulers (not in its tree, version unpinned) - restated here from
the papers (DDPM: Ho et al. 2020 Eq. 7/11, "fixed_small" variance; DDIM: Song et al. 2021 Eq. 12).
Host side only computes the INTEGER timestep table (bit-exact) and three scalars per step; the
per-element update x <- c_x*x + c_eps*eps + c_noise*z runs in the fused HIP sampler kernel
(emo_cfg_step) with counter-based noise, so every rank draws identical z without communication.

Pipeline-enforced config (EMOAnimationPipeline.py:105-130): steps_offset=1 on every scheduler whose config carries the key
(both classes here do, like diffusers' since `timestep_spacing` exists), clip_sample=False.  A scheduler built on its own keeps
diffusers' defaults (DDIM: the pipeline's 1; DDPM: 0) until it is handed to EMOAnimationPipeline.
`timestep_spacing` is "leading" (diffusers' default for both classes: i * (T // n) + steps_offset); other spacings are refused.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch


def _betas(T, beta_start, beta_end, schedule):
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    raise NotImplementedError(f"{schedule} is not implemented")


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                 steps_offset=0, clip_sample=False, set_alpha_to_one=True, timestep_spacing="leading", **_ignored):
        if clip_sample:
            raise ValueError("clip_sample must be False (EMOAnimationPipeline.py:118-130 forces it)")
        if timestep_spacing != "leading":
            raise NotImplementedError(f"timestep_spacing={timestep_spacing!r}: only 'leading' (the diffusers default the reference runs) is built")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, steps_offset=steps_offset, clip_sample=False,
                                      set_alpha_to_one=set_alpha_to_one, timestep_spacing=timestep_spacing)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0).double()
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        self.timesteps = [int(i * ratio) + self.config.steps_offset for i in range(num_inference_steps)][::-1]
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, t, num_inference_steps=None):
        # (a prepared loop state passes ITS step count: the scheduler object is shared, and another prepare_denoise /
        # set_timesteps in between must not change the coefficients of a state that is still being stepped)
        n = self.num_inference_steps if num_inference_steps is None else num_inference_steps
        prev_t = t - self.config.num_train_timesteps // n
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self._final_alpha()
        return a_t, a_prev


class DDIMScheduler(_SchedulerBase):
    """steps_offset defaults to 1 (the pipeline forces it)."""

    def __init__(self, *a, steps_offset=1, eta=0.0, **kw):
        super().__init__(*a, steps_offset=steps_offset, **kw)
        self.eta = eta

    def _final_alpha(self):
        return self.final_alpha_cumprod

    def coefficients(self, t, eta=None, num_inference_steps=None):
        eta = self.eta if eta is None else eta
        a_t, a_prev = self._alphas(t, num_inference_steps)
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * math.sqrt(max(var, 0.0))
        c_x = math.sqrt(a_prev / a_t)
        c_eps = -math.sqrt(a_prev) * math.sqrt(1 - a_t) / math.sqrt(a_t) + math.sqrt(max(1 - a_prev - std * std, 0.0))
        return c_x, c_eps, std


class DDPMScheduler(_SchedulerBase):
    def _final_alpha(self):
        return 1.0

    def coefficients(self, t, eta=None, num_inference_steps=None):
        a_t, a_prev = self._alphas(t, num_inference_steps)
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        k0 = math.sqrt(a_prev) * cur_beta / (1 - a_t)       # coefficient of x0 (Eq. 7)
        kx = math.sqrt(cur_alpha) * (1 - a_prev) / (1 - a_t)  # coefficient of x_t
        c_x = k0 / math.sqrt(a_t) + kx
        c_eps = -k0 * math.sqrt(1 - a_t) / math.sqrt(a_t)
        c_n = math.sqrt(max((1 - a_prev) / (1 - a_t) * cur_beta, 1e-20)) if t > 0 else 0.0
        return c_x, c_eps, c_n
