"""Host logic of the prepared-plan reuse behind EMOAnimationPipeline.__call__ (pipeline.denoise(reuse_state=True), _bind_inputs,
_plan_key) on CPU: the HIP launches are replaced by the oracle-backed stubs of tests/test_dist_gloo.py (test infrastructure).
What is under test: a re-armed plan gives the same latents as a freshly prepared one, for new latents / reference image / text /
guidance scale / seed; plan-changing arguments re-plan; inputs that do not fit the plan are refused."""
import pytest
import torch

from tests import cases
from tests.test_dist_gloo import OracleBackedUNet, _models, _patch_ops


def _pipe():
    from emote_hack_amd import DDIMScheduler
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    _patch_ops()
    unet_sd, ref_sd = _models()
    unet = OracleBackedUNet(cases.TINY_MOTION, unet_sd)
    ref = OracleBackedUNet(cases.TINY, ref_sd, has_out=False)
    return EMOAnimationPipeline(unet=unet, scheduler=DDIMScheduler()), ref


def test_rearmed_plan_equals_fresh_plan():
    from emote_hack_amd.synth import seeded_randn
    pipe, ref = _pipe()
    kw = dict(appearance_encoder=ref, num_inference_steps=2, context_frames=4, context_stride=1, context_overlap=2, reference_group=1)
    a = (seeded_randn((1, 4, 6, 8, 8), 5), seeded_randn((1, 4, 8, 8), 3), seeded_randn((2, 5, 32), 2))
    b = (seeded_randn((1, 4, 6, 8, 8), 15), seeded_randn((1, 4, 8, 8), 13), seeded_randn((2, 5, 32), 12))
    out_a = pipe.denoise(*a, reuse_state=True, guidance_scale=7.5, seed=0, **kw)
    st = pipe._plan_cache[1]
    assert not st.use_graphs                      # no HIP device here: the default is eager
    out_b = pipe.denoise(*b, reuse_state=True, guidance_scale=4.0, seed=1, **kw)
    assert pipe._plan_cache[1] is st              # same plan, inputs re-bound in place
    fresh_b = pipe.denoise(*b, guidance_scale=4.0, seed=1, **kw)
    assert torch.equal(out_b, fresh_b)
    assert torch.equal(pipe.denoise(*a, reuse_state=True, guidance_scale=7.5, seed=0, **kw), out_a)
    assert pipe._plan_cache[1] is st
    assert out_a.data_ptr() != st.latents.data_ptr()   # results are copies: the state's latents belong to the next clip
    # plan-changing arguments prepare afresh: step count, window geometry, guidance on / off
    pipe.denoise(*a, reuse_state=True, guidance_scale=7.5, seed=0, **dict(kw, num_inference_steps=3))
    st3 = pipe._plan_cache[1]
    assert st3 is not st
    pipe.denoise(a[0], a[1], a[2][1:], reuse_state=True, guidance_scale=1.0, seed=0, **dict(kw, num_inference_steps=3))
    assert pipe._plan_cache[1] is not st3 and not pipe._plan_cache[1].cfg


def test_bind_inputs_refuses_what_the_plan_cannot_hold():
    from emote_hack_amd.synth import seeded_randn
    pipe, ref = _pipe()
    kw = dict(appearance_encoder=ref, num_inference_steps=2, context_frames=4, context_stride=1, context_overlap=0)
    lat, refl, text = seeded_randn((1, 4, 4, 8, 8), 5), seeded_randn((1, 4, 8, 8), 3), seeded_randn((2, 5, 32), 2)
    st = pipe.prepare_denoise(lat, refl, text, guidance_scale=7.5, **kw)
    with pytest.raises(ValueError, match="latents"):
        pipe.reset_denoise(st, seeded_randn((1, 4, 5, 8, 8), 5))
    with pytest.raises(ValueError, match="crosses 1.0"):
        pipe.reset_denoise(st, lat, guidance_scale=1.0)
    with pytest.raises(ValueError, match="without audio_features"):
        pipe.reset_denoise(st, lat, audio_features=seeded_randn((4, 3, 32), 7))
    with pytest.raises(ValueError, match="text_embeddings"):
        pipe.reset_denoise(st, lat, text_embeddings=seeded_randn((2, 6, 32), 2))
    pipe.reset_denoise(st, lat, text_embeddings=seeded_randn((2, 5, 32), 9), seed=4)
    assert st.seed == 4 and st.group_ready == -1


def test_cfg_decision_is_made_on_the_f32_value_the_kernel_sees():
    """emo_cfg_step takes guidance_scale as f32: a scale in (1, 1 + 2^-24] is 1.0f there - the host must not plan two planes."""
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    assert EMOAnimationPipeline._do_cfg(7.5) and EMOAnimationPipeline._do_cfg(1.0 + 2.0 ** -20)
    assert not EMOAnimationPipeline._do_cfg(1.0) and not EMOAnimationPipeline._do_cfg(1.0 + 2.0 ** -30) and not EMOAnimationPipeline._do_cfg(0.0)
