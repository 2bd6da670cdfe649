"""GPU parity of the VAE either side of the loop: emote_hack_amd/vae.py AutoencoderKL (HIP kernels) vs
oracle/vae_ref.py (plain torch fp32, the published SD-1.x AutoencoderKL structure; PARITY UNPINNED against diffusers, which is
absent) on name-keyed synthetic weights.  f32 mode at north_star's rtol 1e-3 / atol 1e-4."""
import pytest
import torch

from emote_hack_amd.synth import seeded_randn, synth_state_dict
from tests.test_gpu_unet import check

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_num_groups=8)   # 8x downsampling like SD's
MID = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=2, norm_num_groups=32)


def build(cfg, dtype):
    from emote_hack_amd.vae import AutoencoderKL, vae_param_shapes, VAE_DEFAULTS
    sd = synth_state_dict(vae_param_shapes(dict(VAE_DEFAULTS, **cfg)), prefix="vae.")
    m = AutoencoderKL(**cfg)
    m.load_state_dict(sd)
    return m.to(DEV, dtype), sd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg,hw", [(SMALL, (8, 8)), (MID, (12, 16))])
def test_vae_decode_vs_oracle(dtype, cfg, hw):
    from oracle import vae_ref as V
    m, sd = build(cfg, dtype)
    z = seeded_randn((2, 4, *hw), 7)
    ref = V.decode(sd, z, layers_per_block=cfg["layers_per_block"], groups=cfg["norm_num_groups"])
    got = m.decode(z.to(DEV)).sample
    assert got.shape == ref.shape == (2, 3, 8 * hw[0], 8 * hw[1])
    check(got, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vae_encode_vs_oracle(dtype):
    from oracle import vae_ref as V
    m, sd = build(MID, dtype)
    x = seeded_randn((2, 3, 64, 96), 8).clamp(-1, 1)
    mom = V.encode(sd, x, layers_per_block=2, groups=32)
    dist = m.encode(x.to(DEV)).latent_dist
    assert dist.mean.shape == (2, 4, 8, 12)
    check(dist.mean, mom[:, :4], dtype)
    if dtype == torch.float32:
        torch.testing.assert_close(dist.logvar.cpu(), mom[:, 4:].clamp(-30, 20), rtol=1e-3, atol=1e-4)
        g = torch.Generator(device=DEV).manual_seed(0)
        s = dist.sample(generator=g)
        assert s.shape == dist.mean.shape and bool(torch.isfinite(s).all())


def test_vae_decode_video_is_decode_latents():
    """decode_latents (EMOAnimationPipeline.py:291-307): 1 / 0.18215 scaling, per-frame decode, '(b f) c h w -> b c f h w',
    (video / 2 + 0.5).clamp(0, 1) - frames batched 2 per call here, one per call in the reference: the network is per-image."""
    from oracle import vae_ref as V
    m, sd = build(SMALL, torch.float32)
    lat = 0.2 * seeded_randn((1, 4, 5, 8, 8), 9)
    ref = V.decode_latents(sd, lat, layers_per_block=1, groups=8)
    got = m.decode_video(lat.to(DEV), frames_per_call=2)
    assert got.shape == (1, 3, 5, 64, 64) and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-4)


def test_vae_old_attention_key_names_and_strict_load():
    from emote_hack_amd.vae import AutoencoderKL
    m, sd = build(SMALL, torch.float32)
    old = {}
    for k, v in sd.items():
        for new, o in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            k = k.replace(f".attentions.0.{new}.", f".attentions.0.{o}.")
        old[k] = v[:, :, None, None] if (".attentions.0." in k and k.endswith(".weight") and v.dim() == 2) else v
    m2 = AutoencoderKL(**SMALL)
    assert m2.load_state_dict(old, strict=True) == ([], [])
    m2.to(DEV, torch.float32)
    z = seeded_randn((1, 4, 8, 8), 7)
    assert torch.equal(m2.decode(z.to(DEV)).sample, m.decode(z.to(DEV)).sample)
    with pytest.raises(RuntimeError, match="missing"):
        AutoencoderKL(**SMALL).load_state_dict({k: v for k, v in sd.items() if "conv_out" not in k}, strict=True)


def test_vae_full_size_decode_one_frame_bf16_vs_f32():
    """The SD-1.5 VAE at 512x512 (64x64 latent: 4096 tokens through the single-head mid attention = a 4096 x 4096 score
    matrix between two MFMA GEMMs), one frame: bf16 vs f32 HIP within the bf16 error scale."""
    from emote_hack_amd.vae import AutoencoderKL, vae_param_shapes, VAE_DEFAULTS
    sd = synth_state_dict(vae_param_shapes(VAE_DEFAULTS), prefix="vae.", device=DEV)
    z = 0.5 * seeded_randn((1, 4, 64, 64), 7)
    outs = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = AutoencoderKL()
        m.load_state_dict(sd)
        m.to(DEV, dtype)
        outs[dtype] = m.decode(z.to(DEV)).sample.float().cpu()
        del m
    assert outs[torch.float32].shape == (1, 3, 512, 512) and bool(torch.isfinite(outs[torch.bfloat16]).all())
    check(outs[torch.bfloat16], outs[torch.float32], torch.bfloat16)
