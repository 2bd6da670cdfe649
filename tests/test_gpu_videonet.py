"""GPU parity of the VideoNet path (models/videonet.py:132-267): the HIP ReferenceConditionedAttentionBlock against goldens
produced by the reference's own class bodies, and the whole HIP VideoNet against the oracle (oracle/videonet_ref.py - itself pinned by
those goldens, the wiring golden and the UNet goldens).  f32 at north_star's rtol 1e-3 / atol 1e-4; bf16 within the low-precision
yard-stick."""
import os

import pytest
import torch
from safetensors.torch import load_file

from emote_hack_amd.synth import seeded_randn, synth_state_dict, synth_tensor
from tests import cases
from tests.test_gpu_unet import check

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _videonet(dtype, rcab_block=False):
    from emote_hack_amd.spec import param_shapes
    from emote_hack_amd.videonet import VideoNet
    vn = VideoNet(cases.VIDEONET_TINY, num_frames=4)
    shapes = param_shapes(vn.spec)
    sd = synth_state_dict(shapes, prefix="videonet.")
    if rcab_block:      # the golden block's weights under the first slot: name-keyed by the block's OWN key names, salt 'rcab.'
        slot = "down_blocks.0.attentions.0."
        for k, shp in shapes.items():
            if k.startswith(slot):
                sd[k] = synth_tensor("rcab." + k[len(slot):], shp)
    vn.load_state_dict({"unet." + k: v for k, v in sd.items()})
    return vn.to(DEV, dtype), sd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_conditioned_attention_block_vs_reference_class(dtype):
    """ReferenceConditionedAttentionBlock.forward (models/videonet.py:165-196): sam -> cross_attn -> tam (two temporal transformer
    blocks, no PE), skip_temporal_attn, and the same rows regrouped by update_num_frames - goldens from the reference's class body."""
    g = load_file(os.path.join(cases.GOLDEN_DIR, "videonet.safetensors"))
    vn, _ = _videonet(dtype, rcab_block=True)
    blk = vn.ref_cond_attn_blocks[0]
    x, r, ctx = seeded_randn((8, 64, 4, 8), 83), seeded_randn((8, 64, 4, 8), 84), seeded_randn((8, 5, 32), 85)
    with pytest.raises(Exception, match="update_reference_tensor"):
        blk(x.to(DEV), ctx.to(DEV))
    blk.update_reference_tensor(r.to(DEV))
    check(blk(x.to(DEV), ctx.to(DEV))[0], g["rcab/out"], dtype)
    blk.skip_temporal_attn = True
    check(blk(x.to(DEV), ctx.to(DEV))[0], g["rcab/out_skip"], dtype)
    blk.skip_temporal_attn = False
    blk.update_num_frames(2)
    check(blk(x.to(DEV), ctx.to(DEV))[0], g["rcab/out_frames2"], dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_videonet_forward_vs_oracle(dtype):
    """VideoNet.forward (models/videonet.py:252-267): 2 clips x 4 frames through the whole network - 16 ReferenceConditionedAttentionBlocks
    with their own reference feature maps (64 / 64 / 128 / 128 channels at 16 / 8 / 4 / 2 pixels), per-sample timesteps, and the
    skip_temporal_attn switch - against oracle.videonet_ref.videonet_forward."""
    from oracle import videonet_ref as V
    vn, sd = _videonet(dtype)
    bt, T = 8, 4
    noise = seeded_randn((bt, 4, 16, 16), 90)
    ctx = seeded_randn((bt, 5, 32), 91)
    t = torch.tensor([961, 961, 961, 961, 500, 500, 500, 500])
    geo = {"down_blocks.0": (64, 16), "down_blocks.1": (64, 8), "down_blocks.2": (128, 4), "mid_block": (128, 2),
           "up_blocks.1": (128, 4), "up_blocks.2": (64, 8), "up_blocks.3": (64, 16)}
    refs = [seeded_randn((bt, *[(c, hw, hw) for k, (c, hw) in geo.items() if h.slot.startswith(k + ".")][0]), 100 + i)
            for i, h in enumerate(vn.ref_cond_attn_blocks)]
    with torch.no_grad():
        want = V.videonet_forward(sd, cases.VIDEONET_TINY, noise, t, refs, ctx, T)
        want_skip = V.videonet_forward(sd, cases.VIDEONET_TINY, noise, t, refs, ctx, T, skip_temporal_attn=True)
        # low-precision gate: the same torch statements run in bf16 on the CPU (what the reference's modules would do in that dtype) - this
        # 16-block network is deeper than the tiny motion UNet the relative yard-stick was measured on (torch bf16: mean 1.24e-2 / max 6.4e-2)
        low = low_skip = None
        if dtype != torch.float32:
            b = lambda x_: x_.to(dtype)
            sdl = {k: b(v) for k, v in sd.items()}
            low = V.videonet_forward(sdl, cases.VIDEONET_TINY, b(noise), t, [b(r) for r in refs], b(ctx), T).float()
            low_skip = V.videonet_forward(sdl, cases.VIDEONET_TINY, b(noise), t, [b(r) for r in refs], b(ctx), T, skip_temporal_attn=True).float()
    got = vn(noise.to(DEV), t.to(DEV), [r.to(DEV) for r in refs], ctx.to(DEV))
    assert got.shape == (bt, 4, 16, 16)
    check(got, want, dtype, low)
    check(vn(noise.to(DEV), t.to(DEV), [r.to(DEV) for r in refs], ctx.to(DEV), skip_temporal_attn=True), want_skip, dtype, low_skip)
    assert float((want - want_skip).abs().mean()) > 1e-3          # the temporal modules are live
    if dtype == torch.float32:   # the reference embeddings are live and dealt in order: swapping two of equal shape changes the result
        sw = list(refs)
        sw[0], sw[1] = sw[1], sw[0]
        assert float((vn(noise.to(DEV), t.to(DEV), [r.to(DEV) for r in sw], ctx.to(DEV)).float().cpu() - want).abs().max()) > 1e-3


def test_videonet_copies_the_weights_of_the_unet_it_starts_from():
    """`self.unet = copy.deepcopy(sd_unet)` (models/videonet.py:205): a VideoNet built from a LOADED 2-D UNet holds its tensors under the
    moved key names (attentions.j.* -> attentions.j.cross_attn.*) and becomes usable once sam / tam are loaded."""
    from emote_hack_amd.spec import build_spec, param_shapes
    from emote_hack_amd.unet import UNet3DConditionModel
    from emote_hack_amd.videonet import VideoNet
    u = UNet3DConditionModel(**cases.VIDEONET_TINY)
    usd = synth_state_dict(param_shapes(build_spec(cases.VIDEONET_TINY)), prefix="sd_unet.")
    u.load_state_dict(usd)
    vn = VideoNet(u, num_frames=4)
    k_old, k_new = "down_blocks.0.attentions.0.proj_in.weight", "unet.down_blocks.0.attentions.0.cross_attn.proj_in.weight"
    assert torch.equal(vn._master[k_new[5:]].cpu(), usd[k_old]) and torch.equal(vn._master["conv_in.weight"].cpu(), usd["conv_in.weight"])
    missing = [k for k in vn._shapes if k not in vn._master]
    assert missing and all(".sam." in k or ".tam." in k for k in missing)
    vn.load_state_dict(synth_state_dict({k: vn._shapes[k] for k in missing}, prefix="videonet."), strict=False)
    vn.to(DEV, torch.float32)
    assert vn._w is not None
