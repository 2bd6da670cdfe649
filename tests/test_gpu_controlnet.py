"""GPU parity of the ControlNet branch (magicanimate/models/controlnet.py): the HIP
ControlNetModel vs the CPU oracle (whose in-tree pieces are pinned to the reference by tests/golden/controlnet.safetensors),
and the Backbone consuming its residuals the way the pipeline hands them over (EMOAnimationPipeline.py:514-540)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from emote_hack_amd.spec import build_spec, param_shapes
from emote_hack_amd.synth import seeded_randn, synth_state_dict
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = cases.GOLDEN_DIR
CN_CFG = dict(cases.TINY, down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"))
CN_PREFIX = "controlnet."   # seed salt
_LOOP_CACHE = {}


def build_controlnet(dtype):
    from emote_hack_amd import ControlNetModel
    m = ControlNetModel(**CN_CFG)
    sd = synth_state_dict(param_shapes(m.spec), prefix=CN_PREFIX)
    m.load_state_dict(sd)
    return m.to(DEV, dtype), sd


def yardstick(got, ref, dtype):
    got = got.float().cpu()
    if dtype == torch.float32:
        torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4)
    else:
        k = 1.0 if dtype == torch.bfloat16 else 0.125
        err = (got - ref).abs()
        assert float(err.mean()) < k * 0.03 * float(ref.abs().mean()) + 1e-3, (float(err.mean()), float(ref.abs().mean()))
        assert float(err.max()) < k * 0.10 * float(ref.abs().max()) + 1e-2, (float(err.max()), float(ref.abs().max()))


def test_cond_embedding_vs_reference_golden():
    """ControlNetConditioningEmbedding on the HIP conv kernels vs the reference class's output (f32 mode)."""
    from emote_hack_amd import ControlNetModel
    m = ControlNetModel(**CN_CFG)
    sd = synth_state_dict(param_shapes(m.spec))     # unsalted: the golden used the module's own key names
    m.load_state_dict(sd)
    m.to(DEV, torch.float32)
    g = load_file(os.path.join(G, "controlnet.safetensors"))
    cond = seeded_randn((2, 3, 64, 64), 70)
    rows, h, w = m._cond_embedding(cond.to(DEV), 2, 64, 64)
    assert (h, w) == (8, 8)
    got = rows.float().reshape(2, 8, 8, -1).permute(0, 3, 1, 2).cpu()
    torch.testing.assert_close(got, g["cond_embedding/out"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_controlnet_forward_vs_oracle(dtype):
    from oracle.controlnet_ref import controlnet_forward
    m, sd = build_controlnet(dtype)
    x, ctx = seeded_randn((3, 4, 16, 16), 71), seeded_randn((3, 5, 32), 72)
    cond = seeded_randn((3, 3, 128, 128), 73)
    ref_down, ref_mid = controlnet_forward(sd, CN_CFG, x, 500, ctx, cond, conditioning_scale=0.8)
    out = m(x.to(DEV), 500, ctx.to(DEV), cond.to(DEV), conditioning_scale=0.8)
    assert len(out.down_block_res_samples) == len(ref_down) == 12
    for got, ref in zip(out.down_block_res_samples, ref_down):
        assert tuple(got.shape) == tuple(ref.shape)
        yardstick(got, ref, dtype)
    yardstick(out.mid_block_res_sample, ref_mid, dtype)
    # tuple form + bgr channel order (controlnet.py:465-472)
    down2, mid2 = m(x.to(DEV), 500, ctx.to(DEV), cond.to(DEV), conditioning_scale=0.8, return_dict=False)
    assert torch.equal(mid2, out.mid_block_res_sample)


def test_backbone_consumes_controlnet_residuals():
    """ControlNet (2-D, per frame) -> residuals reshaped '(b f) c h w -> b c f h w' and repeated for CFG
    (EMOAnimationPipeline.py:526-538) -> Backbone forward; HIP f32 vs the oracle chain."""
    from oracle import unet_ref as U
    from oracle.controlnet_ref import controlnet_forward
    from emote_hack_amd import UNet3DConditionModel
    m, sd = build_controlnet(torch.float32)
    Fr = 2
    lat = seeded_randn((1, 4, Fr, 16, 16), 74)
    ctx = seeded_randn((2, 5, 32), 75)
    cond = seeded_randn((Fr, 3, 128, 128), 76)
    frames = lat[0].permute(1, 0, 2, 3)                        # (f, c, h, w)
    ctx_c = ctx[1:].repeat(Fr, 1, 1)
    ref_down, ref_mid = controlnet_forward(sd, CN_CFG, frames, 961, ctx_c, cond)
    out = m(frames.to(DEV), 961, ctx_c.to(DEV), cond.to(DEV))

    def to5(t):                                                # '(b f) c h w -> b c f h w', b = 1, then CFG repeat
        return t.permute(1, 0, 2, 3).unsqueeze(0).repeat(2, 1, 1, 1, 1)
    usd = synth_state_dict(param_shapes(build_spec(cases.TINY_MOTION)))
    unet = UNet3DConditionModel(**cases.TINY_MOTION)
    unet.load_state_dict(usd)
    unet.to(DEV, torch.float32)
    x2 = lat.repeat(2, 1, 1, 1, 1)
    ref = U.unet_forward(usd, cases.TINY_MOTION, x2, 961, ctx, down_block_additional_residuals=[to5(t) for t in ref_down],
                         mid_block_additional_residual=to5(ref_mid))
    y = unet(x2.to(DEV), 961, ctx.to(DEV), down_block_additional_residuals=tuple(to5(t) for t in out.down_block_res_samples),
             mid_block_additional_residual=to5(out.mid_block_res_sample)).sample
    torch.testing.assert_close(y.float().cpu(), ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("graphs", [False, True])
def test_denoise_loop_with_controlnet_vs_oracle(graphs):
    """The sampling loop with the ControlNet branch on (per-frame residual cache -> per-window selection -> Backbone,
    EMOAnimationPipeline.py:514-540,718-746): HIP f32 (eager and HIP-graph replay) vs the oracle loop, 3 DDPM steps,
    8 frames in overlapping windows of 4."""
    from oracle.pipeline_ref import denoise_loop
    from oracle.scheduler_ref import SchedulerRef
    from emote_hack_amd import DDPMScheduler, UNet3DConditionModel
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    cn, cn_sd = build_controlnet(torch.float32)
    usd = synth_state_dict(param_shapes(build_spec(cases.TINY_MOTION)))
    unet = UNet3DConditionModel(**cases.TINY_MOTION)
    unet.load_state_dict(usd)
    unet.to(DEV, torch.float32)
    ref = AppearanceEncoderModel(**cases.TINY)
    rsd = synth_state_dict(param_shapes(ref.spec), prefix=cases.REF_PREFIX)   # the reference-shaped (gutted) ReferenceNet key set
    ref.load_state_dict(rsd)
    ref.to(DEV, torch.float32)
    lat, refl, text = seeded_randn((1, 4, 8, 16, 16), 5), seeded_randn((1, 4, 16, 16), 3), seeded_randn((2, 5, 32), 2)
    cond = seeded_randn((8, 3, 128, 128), 77).clamp(-1, 1) * 0.5 + 0.5
    if "want" not in _LOOP_CACHE:   # the CPU oracle loop is the slow part: once for both parametrisations
        _LOOP_CACHE["want"] = denoise_loop(usd, cases.TINY_MOTION, rsd, cases.TINY, lat, refl, text, scheduler=SchedulerRef("ddpm"),
                                           num_inference_steps=3, guidance_scale=7.5, context_frames=4, context_stride=1,
                                           context_overlap=2, seed=0, controlnet=dict(sd=cn_sd, cfg=CN_CFG, cond=cond, scale=0.9))
        _LOOP_CACHE["base"] = denoise_loop(usd, cases.TINY_MOTION, rsd, cases.TINY, lat, refl, text, scheduler=SchedulerRef("ddpm"),
                                           num_inference_steps=3, guidance_scale=7.5, context_frames=4, context_stride=1,
                                           context_overlap=2, seed=0)
    want = _LOOP_CACHE["want"]
    pipe = EMOAnimationPipeline(unet=unet, controlnet=cn, scheduler=DDPMScheduler())
    got = pipe.denoise(lat.to(DEV), refl, text, appearance_encoder=ref, num_inference_steps=3, guidance_scale=7.5, context_frames=4,
                       context_stride=1, context_overlap=2, seed=0, use_graphs=graphs, controlnet=cn, controlnet_cond=cond,
                       controlnet_conditioning_scale=0.9)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-3, atol=1e-4)
    # and the branch matters: without it the latents differ
    assert float((_LOOP_CACHE["base"] - want).abs().max()) > 1e-2


def test_controlnet_sd15_size_vs_oracle():
    """The SD-1.5-sized ControlNet bench.py --controlnet runs (the Backbone's encoder geometry: 320 / 640 / 1280 / 1280 channels, 8 heads)
    on two 512 x 512 conditioning images (64 x 64 latents), f32 mode, against the oracle: all 12 down residuals and the mid residual at
    north_star's rtol 1e-3 / atol 1e-4 - the full-size plans (halo conv, 256 x 256 tiles, GroupNorm fold, d = 40 / 80 / 160 attention)
    of the (f)1 row, not only the tiny network."""
    from emote_hack_amd import ControlNetModel
    from oracle.controlnet_ref import controlnet_forward
    cfg = dict(cases.SD15, down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",))
    m = ControlNetModel(**cfg)
    sd = synth_state_dict(param_shapes(m.spec), prefix=CN_PREFIX)
    m.load_state_dict(sd)
    m.to(DEV, torch.float32)
    x, ctx = seeded_randn((2, 4, 64, 64), 171), seeded_randn((2, 77, 768), 172)
    cond = seeded_randn((2, 3, 512, 512), 173).clamp(-1, 1) * 0.5 + 0.5
    with torch.no_grad():
        ref_down, ref_mid = controlnet_forward(sd, cfg, x, 500, ctx, cond)
    out = m(x.to(DEV), 500, ctx.to(DEV), cond.to(DEV))
    assert len(out.down_block_res_samples) == len(ref_down) == 12
    for got, ref in zip(out.down_block_res_samples, ref_down):
        torch.testing.assert_close(got.float().cpu(), ref, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out.mid_block_res_sample.float().cpu(), ref_mid, rtol=1e-3, atol=1e-4)
