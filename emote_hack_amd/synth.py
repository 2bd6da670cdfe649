"""Name-keyed synthetic weights.

There is no network for checkpoints and fixtures must stay small, so every state-dict entry is
synthesised *from its key*: `N(0,1)` drawn from `torch.Generator().manual_seed(crc32(key))`,
scaled by the kind of tensor.  The same call reproduces bit-identical weights in the build
container (where they are loaded into the reference model to make golden vectors) and on the
GPU box (where they are loaded into this package's model) - no weights are committed.

  conv / linear weight (ndim >= 2) : N(0,1) / sqrt(fan_in) * gain
  norm weight (1-D '.weight')      : 1 + 0.1 N(0,1)
  bias                             : 0.1 N(0,1)
  '...pos_encoder.pe'              : analytic sinusoid (magicanimate/models/motion_module.py:237-245)

This also un-zeros `motion_modules.*.proj_out` (zero-initialised at motion_module.py:79-80), so
the temporal path is visible to numeric checks.
"""
from __future__ import annotations

import math
import zlib

import torch


def _pe(shape):
    _, max_len, d_model = shape
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def synth_tensor(name: str, shape, gain: float = 1.0, device="cpu") -> torch.Tensor:
    """device='cpu' is the reproducible stream the golden fixtures were made with; a 'cuda' device draws
    from the device generator instead (same distribution, different numbers) - used by bench.py, where
    2.1 G parameters would take a minute on the host."""
    shape = tuple(int(s) for s in shape)
    if name.endswith("pos_encoder.pe"):
        return _pe(shape).to(device)
    g = torch.Generator(device=device).manual_seed(zlib.crc32(name.encode("utf-8")))
    z = torch.randn(shape, generator=g, dtype=torch.float32, device=device)
    if name.endswith(".bias"):
        return 0.1 * z
    if len(shape) == 1:
        return 1.0 + 0.1 * z
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return z * (gain / math.sqrt(fan_in))


def synth_state_dict(shapes: dict, gain: float = 1.0, prefix: str = "", device="cpu") -> dict:
    """shapes: {state_dict_key: shape}.  Returns fp32 CPU tensors keyed by the plain key; `prefix`
    only salts the seed (so a ReferenceNet and a Backbone with equal key names get different
    weights: prefix='reference_unet.')."""
    return {k: synth_tensor(prefix + k, s, gain, device) for k, s in shapes.items()}


def seeded_randn(shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)
