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


def test_text_pairing_by_branch_makes_window_batching_neutral():
    """context_batch_size > 1 with the upstream row pairing (`torch.cat([text] * cbs)`, EMOAnimationPipeline.py:631 against the
    [w0, w1, .., w0, w1, ..] latent rows of :759-763) runs odd windows' uncond rows under the cond text - reproduced literally by
    default (golden `ddim_cbs2`).  text_pairing="branch" is the evident intent: the same function as context_batch_size 1, so the
    two must agree; the literal pairing must not."""
    from emote_hack_amd.synth import seeded_randn
    pipe, ref = _pipe()
    kw = dict(appearance_encoder=ref, num_inference_steps=2, context_frames=4, context_stride=1, context_overlap=2, reference_group=1,
              guidance_scale=7.5, seed=0)
    a = (seeded_randn((1, 4, 8, 8, 8), 5), seeded_randn((1, 4, 8, 8), 3), seeded_randn((2, 5, 32), 2))
    one = pipe.denoise(*a, context_batch_size=1, **kw)
    st = pipe.prepare_denoise(*a, context_batch_size=2, text_pairing="branch", **kw)
    assert all(tv == br for (w, br), tv in st.unit_tv.items()) and st.bank_variants == [1]
    lit = pipe.prepare_denoise(*a, context_batch_size=2, **kw)
    assert any(tv != br for (w, br), tv in lit.unit_tv.items()) and lit.bank_variants == [0, 1]
    two_branch = pipe.denoise(*a, context_batch_size=2, text_pairing="branch", **kw)
    two_literal = pipe.denoise(*a, context_batch_size=2, **kw)
    torch.testing.assert_close(two_branch, one, rtol=1e-3, atol=2e-4)      # (f32 CPU kernels, batches of 4 rows against 2)
    assert (two_literal - one).abs().max() > 1e-3
    with pytest.raises(ValueError, match="text_pairing"):
        pipe.prepare_denoise(*a, text_pairing="both", **kw)


def test_a_plan_keeps_its_own_step_count_and_scheduler():
    """A non-reuse call with another step count between two uses of a cached plan re-times the SHARED scheduler object; the cached
    state must keep stepping with ITS coefficients (its own step count and scheduler object).  A replaced scheduler - same class -
    is another plan."""
    from emote_hack_amd import DDIMScheduler
    from emote_hack_amd.synth import seeded_randn
    pipe, ref = _pipe()
    kw = dict(appearance_encoder=ref, context_frames=4, context_stride=1, context_overlap=0, reference_group=1)
    a = (seeded_randn((1, 4, 4, 8, 8), 5), seeded_randn((1, 4, 8, 8), 3), seeded_randn((2, 5, 32), 2))
    first = pipe.denoise(*a, reuse_state=True, num_inference_steps=4, **kw)
    st = pipe._plan_cache[1]
    pipe.denoise(*a, num_inference_steps=2, **kw)                       # leaves scheduler.num_inference_steps == 2
    assert pipe.scheduler.num_inference_steps == 2
    again = pipe.denoise(*a, reuse_state=True, num_inference_steps=4, **kw)
    assert pipe._plan_cache[1] is st and torch.equal(again, first)
    # a state stepped by hand after the scheduler moved on
    st2 = pipe.prepare_denoise(*a, num_inference_steps=4, **kw)
    pipe.prepare_denoise(*a, num_inference_steps=2, **kw)
    assert torch.equal(pipe._run_loop(st2), first)
    pipe.scheduler = DDIMScheduler()                                     # same class, new object: not the cached plan's scheduler
    pipe.denoise(*a, reuse_state=True, num_inference_steps=4, **kw)
    assert pipe._plan_cache[1] is not st
    pipe.clear_plan_cache()
    assert pipe._plan_cache is None


def test_omitted_guidance_eta_seed_mean_their_defaults_on_a_cache_hit():
    from emote_hack_amd.synth import seeded_randn
    pipe, ref = _pipe()
    kw = dict(appearance_encoder=ref, num_inference_steps=2, context_frames=4, context_stride=1, context_overlap=0, reference_group=1)
    a = (seeded_randn((1, 4, 4, 8, 8), 5), seeded_randn((1, 4, 8, 8), 3), seeded_randn((2, 5, 32), 2))
    fresh = pipe.denoise(*a, **kw)                                       # guidance 7.5, eta 0, seed 0
    pipe.denoise(*a, reuse_state=True, guidance_scale=3.0, seed=5, eta=0.5, **kw)
    st = pipe._plan_cache[1]
    out = pipe.denoise(*a, reuse_state=True, **kw)
    assert pipe._plan_cache[1] is st and (st.guidance_scale, st.eta, st.seed) == (7.5, 0.0, 0)
    assert torch.equal(out, fresh)


def test_emulated_rank_runs_its_units_and_its_share_of_the_reference_groups():
    """prepare_denoise(_emulate_rank=(r, n)) - bench.py --emulate-rank: rank r's units of an n-rank job in one process, no
    collectives; its eps slices equal what the full single-process run produces for those units in the first step."""
    from emote_hack_amd.synth import seeded_randn
    pipe, ref = _pipe()
    kw = dict(appearance_encoder=ref, num_inference_steps=2, context_frames=4, context_stride=1, context_overlap=0, reference_group=2, return_eps=True)
    a = (seeded_randn((1, 4, 8, 8, 8), 5), seeded_randn((1, 4, 8, 8), 3), seeded_randn((2, 5, 32), 2))
    full = pipe.prepare_denoise(*a, **kw)
    assert full.units == [(0, 0), (1, 0), (0, 1), (1, 1)]
    pipe.denoise_step(full, 0)
    for r in range(4):
        st = pipe.prepare_denoise(*a, _emulate_rank=(r, 4), **kw)
        assert st.units == [full.units[r]] and st.world_size == 4 and not st.dist and st.T == 2 and not st.lookahead      # (no HIP device here: no side stream)
        pipe.denoise_step(st, 0)
        assert bool(torch.isfinite(st.latents).all())
        i = full.units.index(st.units[0])
        if st.units[0][1] == 0:     # an uncond unit reads no bank: the same forward (batched with its sibling there, alone here)
            torch.testing.assert_close(st.send[0], full.send[i], rtol=1e-4, atol=1e-5)
        else:                       # a cond unit reads the stand-in for the gathered banks (this rank's timesteps, repeated)
            assert bool(torch.isfinite(st.send[0]).all())
    with pytest.raises(ValueError):
        pipe.prepare_denoise(*a, _emulate_rank=(0, 2), dist=True, **kw)
