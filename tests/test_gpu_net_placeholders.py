"""GPU parity of Net.py's placeholder modules (emote_hack_amd/net_placeholders.py) against goldens produced by the reference's
own class bodies (tools/oracle/gen_golden.py gen_net_placeholders), and the same refusals where the reference raises
(tests/golden/net_placeholders.json)."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from emote_hack_amd.synth import seeded_randn, synth_state_dict
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {torch.float32: dict(rtol=1e-3, atol=1e-4), torch.bfloat16: dict(rtol=5e-2, atol=5e-2)}


@pytest.fixture(scope="module")
def g():
    return load_file(os.path.join(cases.GOLDEN_DIR, "net_placeholders.safetensors"))


def _mk(mod, prefix, dtype):
    mod.load_state_dict(synth_state_dict(mod._shapes, prefix=prefix))
    return mod.to(DEV, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_attention_and_motion_module_vs_reference(g, dtype):
    from emote_hack_amd.net_placeholders import MotionModule, ReferenceAttention
    ra = _mk(ReferenceAttention(64), "net_reference_attention.", dtype)
    y = ra(seeded_randn((2, 64, 4, 6), 400).to(DEV), seeded_randn((2, 64, 4, 6), 401).to(DEV))
    torch.testing.assert_close(y.float().cpu(), g["reference_attention/out"], **TOL[dtype])
    mm = _mk(MotionModule(64, 3), "net_motion_module.", dtype)
    y = mm(seeded_randn((2, 64, 6, 1, 1), 402).to(DEV))
    torch.testing.assert_close(y.float().cpu(), g["motion_module/out"], **TOL[dtype])


def test_the_geometries_the_reference_rejects_are_rejected():
    from emote_hack_amd.net_placeholders import BackboneNetwork, MotionModule, TemporalModule
    J = json.load(open(os.path.join(cases.GOLDEN_DIR, "net_placeholders.json")))
    assert J["motion_module_4x4"] == J["motion_module_even_kernel"] == J["temporal_module"] == "RuntimeError"
    mm = _mk(MotionModule(64, 3), "net_motion_module.", torch.float32)
    with pytest.raises(ValueError, match="1x1"):
        mm(seeded_randn((2, 64, 6, 4, 4), 403).to(DEV))
    with pytest.raises(ValueError, match="even"):
        _mk(MotionModule(64, 4), "net_motion_module4.", torch.float32)(seeded_randn((2, 64, 6, 1, 1), 402).to(DEV))
    with pytest.raises(NotImplementedError):
        TemporalModule(64, 4)(seeded_randn((2, 64, 4, 4), 404), None)
    assert J["backbone_forward"] == "AssertionError"
    assert BackboneNetwork is not None


def test_backbone_network_runs_what_the_reference_runs(g):
    """BackboneNetwork.forward (Net.py:397-411): the reference- and audio-attention stacks equal the reference's; the temporal stage
    trips the ndim == 5 assertion there and here."""
    from emote_hack_amd.conditioning import AudioAttentionLayers
    from emote_hack_amd.net_placeholders import BackboneNetwork
    feat = 32
    audio = AudioAttentionLayers(feat, 2)
    audio.load_state_dict(synth_state_dict(audio._shapes, prefix="net_backbone_audio."))
    audio.to(DEV, torch.float32)
    bb = BackboneNetwork(feat, 2, lambda img: img, audio, temporal_module_kwargs=dict(num_attention_heads=4, num_transformer_block=1))
    shp = {f"{i}.{n}.{w}": ((feat, feat) if w == "weight" else (feat,)) for i in range(2) for n in ("query", "key", "value") for w in ("weight", "bias")}
    bb.load_state_dict({"reference_attention_layers." + k: v for k, v in synth_state_dict(shp, prefix="net_backbone_ref.").items()})
    bb.to(DEV, torch.float32)
    lat, aud, ref = seeded_randn((2, 5, feat), 405), seeded_randn((2, 5, feat), 406), seeded_randn((2, 1, feat), 407)
    y = bb.forward_before_temporal(lat.to(DEV), aud.to(DEV), ref.to(DEV))
    torch.testing.assert_close(y.float().cpu(), g["backbone/before_temporal"], rtol=1e-3, atol=1e-4)
    with pytest.raises(AssertionError, match="ndim=5"):
        bb(lat.to(DEV), aud.to(DEV), ref.to(DEV))
