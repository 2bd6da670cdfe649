"""Oracle (test infrastructure): DDPM / DDIM scheduler + context windows + counter-based noise.

PARITY UNPINNED for the scheduler: the reference calls `diffusers` schedulers (not in tree,
version unpinned - requirements.txt:1; call sites EMOAnimationPipeline.py:653-654,764,817).
Restated from the papers: DDPM = Ho et al. 2020 Eq. 7 / Eq. 11 with "fixed_small" variance,
DDIM = Song et al. 2021 Eq. 12 (eta).  In-tree pins honoured: steps_offset=1 and
clip_sample=False are forced by the pipeline ctor on EVERY scheduler whose config carries the key
(EMOAnimationPipeline.py:105-130: `hasattr(scheduler.config, "steps_offset")` - DDIM always, DDPM in
every diffusers release that has `timestep_spacing`), so SchedulerRef defaults to offset 1 for both
kinds: 50 steps of 1000 run the INT table [981, 961, ..., 1]; the
x0-reconstruction identity of `next_step` (EMOAnimationPipeline.py:379-400) is a self-check.

`uniform_windows` restates magicanimate/pipelines/context.py:12-42 and IS pinned (integer
goldens in tests/golden/ints.json).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def make_betas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear"):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


def timestep_table(num_inference_steps, num_train_timesteps=1000, steps_offset=0):
    """INT, bit-exact: 'leading' spacing (diffusers' default `timestep_spacing`): i * (T // n) + steps_offset, descending.
    50 on 1000 -> [980..0] at offset 0, [981..1] at the offset 1 the pipeline forces."""
    ratio = num_train_timesteps // num_inference_steps
    return [int(i * ratio) + steps_offset for i in range(num_inference_steps)][::-1]


class SchedulerRef:
    def __init__(self, kind="ddim", num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="linear", steps_offset=1, set_alpha_to_one=True, eta=0.0):
        assert kind in ("ddim", "ddpm")
        self.kind, self.T, self.eta = kind, num_train_timesteps, eta
        self.steps_offset = steps_offset
        self.betas = make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = timestep_table(n, self.T, self.steps_offset)
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x, noise=None):
        prev_t = t - self.T // self.n
        a_t = self.alphas_cumprod[t].double()
        a_prev = (self.alphas_cumprod[prev_t] if prev_t >= 0 else
                  (self.final_alpha_cumprod if self.kind == "ddim" else torch.tensor(1.0))).double()
        x, eps = x.double(), eps.double()
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        if self.kind == "ddim":
            var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
            std = self.eta * var.sqrt()
            out = a_prev.sqrt() * x0 + (1 - a_prev - std ** 2).sqrt() * eps
            if self.eta > 0:
                out = out + std * noise.double()
        else:
            cur_alpha = a_t / a_prev
            cur_beta = 1 - cur_alpha
            out = (a_prev.sqrt() * cur_beta / (1 - a_t)) * x0 + (cur_alpha.sqrt() * (1 - a_prev) / (1 - a_t)) * x
            if t > 0:
                var = ((1 - a_prev) / (1 - a_t) * cur_beta).clamp(min=1e-20)
                out = out + var.sqrt() * noise.double()
        return out.float()

    def coefficients(self, t):
        """Per-step scalars (c_x, c_eps, c_noise) such that x_prev = c_x*x + c_eps*eps + c_noise*z.
        This is the form the fused HIP cfg_step kernel consumes."""
        prev_t = t - self.T // self.n
        a_t = float(self.alphas_cumprod[t].double())
        if prev_t >= 0:
            a_prev = float(self.alphas_cumprod[prev_t].double())
        else:
            a_prev = float(self.final_alpha_cumprod.double()) if self.kind == "ddim" else 1.0
        if self.kind == "ddim":
            var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
            std = self.eta * math.sqrt(max(var, 0.0))
            c_x = math.sqrt(a_prev / a_t)
            c_eps = -math.sqrt(a_prev) * math.sqrt(1 - a_t) / math.sqrt(a_t) + math.sqrt(max(1 - a_prev - std * std, 0.0))
            return c_x, c_eps, std
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        k0 = math.sqrt(a_prev) * cur_beta / (1 - a_t)
        kx = math.sqrt(cur_alpha) * (1 - a_prev) / (1 - a_t)
        c_x = k0 / math.sqrt(a_t) + kx
        c_eps = -k0 * math.sqrt(1 - a_t) / math.sqrt(a_t)
        c_n = math.sqrt(max((1 - a_prev) / (1 - a_t) * cur_beta, 1e-20)) if t > 0 else 0.0
        return c_x, c_eps, c_n


# ----------------------------------------------------------------------------- windows (INT)

def ordered_halving(val: int) -> float:
    """context.py:12-17: bit-reverse a 64-bit integer and read it as a fraction in [0,1)."""
    rev = 0
    for i in range(64):
        if (val >> i) & 1:
            rev |= 1 << (63 - i)
    return rev / (1 << 64)


def uniform_windows(step, num_steps, num_frames, context_size, context_stride=3, context_overlap=4,
                    closed_loop=True):
    """context.py:20-42 (uniform).  Returns list[list[int]]."""
    if num_frames <= context_size:
        return [list(range(num_frames))]
    out = []
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for k in range(context_stride):
        cstep = 1 << k
        pad = int(round(num_frames * ordered_halving(step)))
        start = int(ordered_halving(step) * cstep) + pad
        stop = num_frames + pad + (0 if closed_loop else -context_overlap)
        stride = context_size * cstep - context_overlap
        for j in range(start, stop, stride):
            out.append([e % num_frames for e in range(j, j + context_size * cstep, cstep)])
    return out


# ----------------------------------------------------------------------------- counter-based noise

def _mix32(x):
    x = np.asarray(x, dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & m
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & m
    x = (x ^ (x >> np.uint64(16))) & m
    return x


def counter_normal(seed: int, step: int, n: int) -> torch.Tensor:
    """Counter-based N(0,1): element i of step s draws two 32-bit hashes of (seed, s, i) and
    applies Box-Muller.  Identical on every rank without communication (SURVEY.md 8e
    'Determinism').  The integer hash is bit-exact with emo_cfg_step's device code; the float
    transform agrees to ~1e-6."""
    idx = np.arange(n, dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    key = _mix32((np.uint64(seed) & m) ^ (_mix32(np.uint64(step) + np.uint64(0x9E3779B9)) ))
    h1 = _mix32((idx * np.uint64(2) + np.uint64(0)) & m ^ key)
    h2 = _mix32((idx * np.uint64(2) + np.uint64(1)) & m ^ key)
    u1 = (h1.astype(np.float64) + 1.0) / 4294967296.0          # (0, 1]
    u2 = h2.astype(np.float64) / 4294967296.0                  # [0, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32))
