"""GPU parity for the EMO conditioning modules vs goldens captured from the
reference's AST-extracted classes (tests/golden/conditioning.safetensors).  f32 mode, rtol 1e-3 / atol 1e-4;
the speed-bucket index is INT bit-exact."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from emote_hack_amd.synth import seeded_randn, synth_state_dict
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def g():
    return load_file(os.path.join(cases.GOLDEN_DIR, "conditioning.safetensors"))


def mk(cls, prefix, *a, dtype=torch.float32):
    m = cls(*a)
    m.load_state_dict(synth_state_dict(m.state_dict_shapes(), prefix=prefix))
    return m.to(DEV, dtype)


def close(a, b, rtol=1e-3, atol=1e-4):
    torch.testing.assert_close(a.float().cpu(), b, rtol=rtol, atol=atol)


def test_speed_modules(g):
    from emote_hack_amd.conditioning import SpeedController, SpeedEncoder
    sp = torch.tensor(cases.SPEEDS, dtype=torch.float32)
    se = mk(SpeedEncoder, "speed_encoder.", 9, 64)
    close(se.encode_speed(sp), g["speed_encoder/encode"], 1e-4, 1e-5)
    close(se(sp), g["speed_encoder/out"])
    with pytest.raises(AssertionError):
        SpeedEncoder(10, 64)   # the reference ctor asserts the same way (EMOAnimationPipeline.py:165-167 is broken upstream)
    sc = mk(SpeedController, "speed_controller.", 9, 64)
    ints = json.load(open(os.path.join(cases.GOLDEN_DIR, "ints.json")))
    assert sc.map_speed_to_bucket(sp).cpu().tolist() == ints["speed_buckets"]["buckets"]   # INT bit-exact
    close(sc(sp), g["speed_controller/out"])


def test_face_region_controller(g):
    from emote_hack_amd.conditioning import FaceRegionController
    fr = mk(FaceRegionController, "face_region.", 1, 32)
    close(fr(seeded_randn((2, 1, 8, 8), 60)), g["face_region/out"])


def test_net_attention_layers(g):
    from emote_hack_amd.conditioning import AudioAttentionLayers, ReferenceAttentionLayer
    al = mk(AudioAttentionLayers, "audio_layers.", 48, 2)
    close(al(seeded_randn((2, 6, 48), 61), seeded_randn((2, 6, 48), 62)), g["audio_layers/out"])
    rl = mk(ReferenceAttentionLayer, "ref_layer.", 48)
    close(rl(seeded_randn((2, 6, 48), 63), seeded_randn((2, 1, 48), 64)), g["ref_layer/out"])


def test_stage2_attentions(g):
    from emote_hack_amd.conditioning import AudioAttention, TemporalAttention
    aa = mk(AudioAttention, "stage2_audio.", 64, 768, 8)
    close(aa(seeded_randn((2, 12, 64), 65), seeded_randn((2, 5, 768), 66)), g["stage2_audio/out"])
    ta = mk(TemporalAttention, "stage2_temporal.", 64, 8)
    close(ta(seeded_randn((2, 12, 64), 67)), g["stage2_temporal/out"])
    aab = mk(AudioAttention, "stage2_audio.", 64, 768, 8, dtype=torch.bfloat16)
    torch.testing.assert_close(aab(seeded_randn((2, 12, 64), 65), seeded_randn((2, 5, 768), 66)).cpu(), g["stage2_audio/out"], rtol=5e-2, atol=5e-2)


def test_stage3_combine_rule():
    from emote_hack_amd.conditioning import stage3_combine
    lat, face, spd = seeded_randn((2, 4, 8, 8), 70), seeded_randn((2, 4, 8, 8), 71), seeded_randn((2, 4), 72)
    out = stage3_combine(lambda x: x * 1.0, lat.to(DEV), face.to(DEV), spd.to(DEV))
    torch.testing.assert_close(out.cpu(), lat + face + spd[:, :, None, None], rtol=1e-5, atol=1e-6)


def test_unet_accepts_audio_and_speed_kwargs():
    """The EMO extension of the UNet boundary (EMOAnimationPipeline.py:783-784): per-frame audio context replaces the
    text context in attn2, speed embeddings are added to the time embedding.  HIP vs the oracle's statement."""
    from emote_hack_amd.spec import build_spec, param_shapes
    from emote_hack_amd.unet import UNet3DConditionModel
    from oracle import unet_ref as U
    cfg = cases.TINY_MOTION
    sd = synth_state_dict(param_shapes(build_spec(cfg)))
    x, ctx = cases.tiny_inputs(2, 4)
    audio = seeded_randn((8, 5, 32), 80)      # (B*F, L_a, D)
    speed = 0.1 * seeded_randn((2, 128), 81)  # (B, 4*C0)
    ref = U.unet_forward(sd, cfg, x, 500, ctx, audio_features=audio, speed_embeddings=speed)
    m = UNet3DConditionModel(**cfg)
    m.load_state_dict(sd)
    m.to(DEV, torch.float32)
    y = m(x.to(DEV), 500, ctx.to(DEV), audio_features=audio.to(DEV), speed_embeddings=speed.to(DEV)).sample
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_audio_windows_vs_reference_loop(dtype):
    """Net.py:649-667 (the per-frame windowing loop of Wav2VecFeatureExtractor, run by tools/oracle/gen_golden.py on synthetic
    hidden states): bit-exact - the kernel is a gather with zero padding."""
    from safetensors.torch import load_file
    from emote_hack_amd.conditioning import audio_context_tokens, audio_windows
    g = load_file(os.path.join(cases.GOLDEN_DIR, "audio_windows.safetensors"))
    for name, (m, n) in {"t9_m2n2": (2, 2), "t3_m2n2": (2, 2), "t7_m1n3": (1, 3), "t5_m0n0": (0, 0)}.items():
        x = g[f"{name}/in"].to(dtype)
        got = audio_windows(x.to(DEV).unsqueeze(0), m, n)
        want = x.float()
        ref = g[f"{name}/out"].to(dtype)
        assert torch.equal(got.cpu(), ref), name
        assert want.shape[0] == got.shape[0]
    tok = audio_context_tokens(audio_windows(g["t9_m2n2/in"].to(DEV), 2, 2), 4, feature_dim=16)
    assert tok.shape == (4, 5, 16) and torch.equal(tok[0, 2].cpu(), g["t9_m2n2/in"][0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_videonet_attention_modules_vs_reference(dtype):
    """SURVEY A19 (models/videonet.py:15-128): the width-concat reference attention and the per-pixel temporal attention vs
    goldens produced by the reference's own classes (AST-extracted; xformers' memory_efficient_attention stood in by its
    published softmax(q k^T K^-0.5) v semantics)."""
    from emote_hack_amd.conditioning import SpatialAttentionModule, TemporalAttentionModule
    g = load_file(os.path.join(cases.GOLDEN_DIR, "videonet.safetensors"))
    tol = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    sp = mk(SpatialAttentionModule, "videonet_spatial.", 64, 64, 8).to(DEV, dtype)
    y = sp(seeded_randn((3, 64, 4, 8), 80).to(DEV), seeded_randn((3, 64, 4, 8), 81).to(DEV))
    torch.testing.assert_close(y.float().cpu(), g["spatial/out"], **tol)
    tm = mk(TemporalAttentionModule, "videonet_temporal.", 64, 4, 64, 8).to(DEV, dtype)
    y = tm(seeded_randn((2 * 4, 64, 4, 4), 82).to(DEV))
    torch.testing.assert_close(y.float().cpu(), g["temporal/out"], **tol)
    with pytest.raises(ValueError):
        SpatialAttentionModule(64)          # the reference default embed_dim=40 cannot add attn_out to 64-channel tokens


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_face_locator_vs_reference(g, dtype):
    """SURVEY 8f row 4, Net.py:819-855: three conv3x3 + ReLU + 2x2 max-pool stages, a 1x1 conv and the bilinear upsampling of the
    logits (align_corners=False) - vs the golden of the reference's own class."""
    from emote_hack_amd.conditioning import FaceLocator
    fl = mk(FaceLocator, "face_locator.").to(DEV, dtype)
    y = fl(seeded_randn((2, 3, 32, 48), 68).to(DEV))
    assert y.shape == (2, 1, 32, 48)
    tol = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(y.cpu(), g["face_locator/out"], **tol)
    with pytest.raises(AssertionError):
        fl(seeded_randn((2, 3, 32, 48), 68).to(DEV).half())
