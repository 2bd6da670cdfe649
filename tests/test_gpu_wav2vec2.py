"""GPU parity of the audio front-end (Net.py:607-667): the HIP Wav2Vec2Model / Wav2VecFeatureExtractor against the outputs
of transformers' own Wav2Vec2Model on the same name-keyed weights and seeded waveforms (tests/golden/wav2vec2.safetensors,
tools/oracle/gen_golden_wav2vec2.py).  f32 mode at north_star's rtol 1e-3 / atol 1e-4; bf16 / fp16 against the same goldens at the
kernel-level low-precision tolerance scaled by the 12 post-LN layers."""
import os

import pytest
import torch
from safetensors.torch import load_file

from emote_hack_amd.synth import seeded_randn
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def gold():
    return load_file(os.path.join(cases.GOLDEN_DIR, "wav2vec2.safetensors"))


def _model(cfg, dtype):
    from emote_hack_amd.wav2vec2 import Wav2Vec2Model, wav2vec2_synth_state_dict
    m = Wav2Vec2Model(cfg)
    m.load_state_dict(wav2vec2_synth_state_dict(cfg))
    return m.to(DEV, dtype)


def test_new_frontend_kernels_vs_torch():
    """emo_channelnorm (GroupNorm(C, C) over a sequence, + GELU) and emo_act's erf-GELU against torch."""
    import torch.nn.functional as F
    from emote_hack_amd import ops
    x = seeded_randn((799, 96), 7) * 2 + 0.3
    g, b = 1 + 0.1 * seeded_randn((96,), 8), 0.1 * seeded_randn((96,), 9)
    for gelu in (False, True):
        ref = F.group_norm(x.t()[None], 96, g, b, 1e-5)[0].t()
        ref = F.gelu(ref) if gelu else ref
        got = ops.channel_norm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, gelu=gelu)
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(ops.act((3 * x).to(DEV), "gelu").cpu(), F.gelu(3 * x), rtol=1e-3, atol=1e-4)
    xb = x.bfloat16()
    torch.testing.assert_close(ops.act(xb.to(DEV), "gelu").float().cpu(), F.gelu(xb.float()), rtol=2e-2, atol=2e-2)


def test_wav2vec2_tiny_f32(gold):
    y = _model(cases.WAV2VEC2_TINY, torch.float32)(0.5 * seeded_randn((1, 4000), 501)).last_hidden_state
    torch.testing.assert_close(y.cpu(), gold["tiny/out"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_wav2vec2_base_vs_transformers_golden(gold, dtype):
    """The wav2vec2-base configuration (7 conv layers, positional conv k = 128 in 16 groups, 12 layers of 12 heads x 64) on 1 s of
    audio -> (1, 49, 768), against transformers' output."""
    y = _model({}, dtype)(gold["base/input_values"]).last_hidden_state.cpu()
    assert y.shape == (1, 49, 768) and y.dtype == torch.float32
    if dtype == torch.float32:
        torch.testing.assert_close(y, gold["base/out"], rtol=1e-3, atol=1e-4)
    else:
        e = (y - gold["base/out"]).abs()
        tol_mean, tol_max = (3e-2, 0.5) if dtype == torch.bfloat16 else (4e-3, 8e-2)     # LayerNorm outputs of unit scale
        print(f"wav2vec2-base {dtype}: mean err {float(e.mean()):.3e} max {float(e.max()):.3e}")
        assert float(e.mean()) < tol_mean and float(e.max()) < tol_max


def test_feature_extractor_matches_the_reference_front_end(gold):
    """Wav2VecFeatureExtractor.extract_features(waveform): processor normalisation -> encoder -> the reference's windowing
    (Net.py:636-667) = golden `base/features` (transformers model + the reference's own windowing statements)."""
    from emote_hack_amd.wav2vec2 import Wav2VecFeatureExtractor, normalize_waveform
    wave = 0.1 * seeded_randn((16000,), 502) + 0.05 * torch.sin(torch.arange(16000) * 0.05)
    torch.testing.assert_close(normalize_waveform(wave), gold["base/input_values"], rtol=1e-5, atol=1e-6)
    fx = Wav2VecFeatureExtractor(_model({}, torch.float32), DEV)
    feats = fx.extract_features(wave, m=2, n=2)
    assert feats.shape == (49, 5 * 768)
    torch.testing.assert_close(feats.float().cpu(), gold["base/features"], rtol=1e-3, atol=1e-4)
    # stereo input: mean over channels first (Net.py:634-636)
    st = torch.stack([wave * 1.5, wave * 0.5], 1)
    torch.testing.assert_close(fx.extract_features(st).float().cpu(), gold["base/features"], rtol=1e-3, atol=1e-4)


def test_pipeline_call_accepts_raw_audio(gold):
    """EMOAnimationPipeline.__call__(audio=waveform, feature_extractor=...) (EMOAnimationPipeline.py:592-593: `audio_features =
    feature_extractor.extract_features_from_mp4(audio, m=2, n=2)`): the features are computed by the HIP front-end and reach the
    UNet as the per-frame attn2 context - same latents as passing audio_features= explicitly."""
    from emote_hack_amd import DDIMScheduler
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.conditioning import audio_context_tokens
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    from emote_hack_amd.wav2vec2 import Wav2VecFeatureExtractor
    from tests.test_gpu_unet import build
    cfg = dict(cases.TINY_MOTION, cross_attention_dim=64)
    unet = build(cfg, torch.float32)
    ref = build(dict(cases.TINY, cross_attention_dim=64), torch.float32, cases.REF_PREFIX, cls=AppearanceEncoderModel, has_out=False)
    fx = Wav2VecFeatureExtractor(_model(cases.WAV2VEC2_TINY, torch.float32), DEV)
    wave = 0.5 * seeded_randn((4000,), 501)
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDIMScheduler())
    kw = dict(video_length=4, height=128, width=128, num_inference_steps=2, guidance_scale=7.5, context_frames=4, context_stride=1,
              context_overlap=0, output_type="latent", appearance_encoder=ref, text_embeddings=seeded_randn((2, 5, 64), 2),
              ref_image_latents=seeded_randn((1, 4, 16, 16), 3), latents=seeded_randn((1, 4, 4, 16, 16), 1).to(DEV), seed=0)
    a = pipe("", audio=wave, feature_extractor=fx, **kw).videos
    feats = audio_context_tokens(fx.extract_features(wave), 4, 64)
    assert feats.shape == (4, 5, 64)
    b = pipe("", audio_features=feats, **kw).videos
    assert torch.equal(a, b)
    c = pipe("", **kw).videos
    assert float((a - c).abs().max()) > 1e-4          # the audio context is live
    with pytest.raises(ValueError, match="feature_extractor"):
        pipe("", audio=wave, **kw)
