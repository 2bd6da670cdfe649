#!/usr/bin/env python3
"""Per-step error of the HIP sampling loop (f32 mode) against the reference-generated loop goldens
(tests/golden/loop_tiny.safetensors): where does the 2e-3 / 2e-4 tolerance of tests/test_gpu_unet.py come from?
Prints, per step, max |eps - golden|, the largest violation ratio |d| / (atol + rtol |ref|) at north_star's 1e-3 / 1e-4, and the
same for the two CFG branches' contribution (the guidance formula eps = uc + 7.5 (c - uc) amplifies the per-branch error)."""
import os
import sys

import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from emote_hack_amd import DDIMScheduler, DDPMScheduler  # noqa: E402
from emote_hack_amd.appearance_encoder import AppearanceEncoderModel  # noqa: E402
from emote_hack_amd.pipeline import EMOAnimationPipeline  # noqa: E402
from emote_hack_amd.spec import param_shapes  # noqa: E402
from emote_hack_amd.synth import seeded_randn, synth_state_dict  # noqa: E402
from emote_hack_amd.unet import UNet3DConditionModel  # noqa: E402
from tests import cases  # noqa: E402

DEV = "cuda"


def build(cls, cfg, prefix=""):
    m = cls(**cfg)
    m.load_state_dict(synth_state_dict(param_shapes(m.spec), prefix=prefix))
    return m.to(DEV, torch.float32)


g = load_file(os.path.join(cases.GOLDEN_DIR, "loop_tiny.safetensors"))
ref = build(AppearanceEncoderModel, cases.TINY, cases.REF_PREFIX)
unet = build(UNet3DConditionModel, cases.TINY_MOTION)
for kind in ("ddim", "ddpm"):
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDIMScheduler() if kind == "ddim" else DDPMScheduler())
    lat, eps = pipe.denoise(seeded_randn((1, 4, 8, 16, 16), 5).to(DEV), seeded_randn((1, 4, 16, 16), 3), seeded_randn((2, 5, 32), 2),
                            appearance_encoder=ref, num_inference_steps=3, guidance_scale=7.5, context_frames=4, context_stride=1,
                            context_overlap=2, seed=0, return_eps=True)
    for i in range(3):
        a, b = eps[i].cpu(), g[f"{kind}/eps{i}"]
        d = (a - b).abs()
        print(f"{kind} step {i}: max|d eps| {float(d.max()):.2e}  mean {float(d.mean()):.2e}  max |ref| {float(b.abs().max()):.2f}  "
              f"worst d/(1e-4+1e-3|ref|) {float((d / (1e-4 + 1e-3 * b.abs())).max()):.2f}")
    d = (lat.cpu() - g[f"{kind}/latents"]).abs()
    print(f"{kind} latents: max|d| {float(d.max()):.2e}  worst d/(1e-4+1e-3|ref|) {float((d / (1e-4 + 1e-3 * g[f'{kind}/latents'].abs())).max()):.2f}")
