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


def test_vae_full_size_decode_one_frame_vs_oracle_and_bf16_vs_f32():
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
    # ... and the f32 mode against the ORACLE at this size (one 512 x 512 frame of the SD-1.x decoder: ~10 s of CPU time)
    from oracle import vae_ref as V
    with torch.no_grad():
        ref = V.decode({k: v.float().cpu() for k, v in sd.items()}, z)
    torch.testing.assert_close(outs[torch.float32], ref, rtol=1e-3, atol=1e-4)


def test_pipeline_source_image_goes_through_images2latents(tmp_path):
    """EMOAnimationPipeline.py:686-689 + 402-414: `source_image` is a file path (opened, resized to (width, height)) or an
    (H, W, 3) uint8 array; images2latents scales by / 127.5 - 1, moves channels first and takes `vae.encode(x)['latent_dist'].mean
    * 0.18215`.  A reference-style call with vae= + source_image= must produce the same video as one fed the oracle's latents
    of that image (the path used to hand the raw HWC uint8 array to vae.encode)."""
    import numpy as np
    from PIL import Image
    from oracle import vae_ref as V
    from emote_hack_amd import DDIMScheduler
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    from tests import cases
    from tests.test_gpu_unet import build as build_unet
    vae, vsd = build(SMALL, torch.float32)
    ref = build_unet(cases.TINY, torch.float32, cases.REF_PREFIX, cls=AppearanceEncoderModel, has_out=False)
    unet = build_unet(cases.TINY_MOTION, torch.float32)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(128, 128, 3)).astype(np.uint8)
    want_lat = V.encode(vsd, torch.from_numpy(img).float().div(127.5).sub(1.0).permute(2, 0, 1)[None], layers_per_block=1, groups=8)[:, :4] * 0.18215
    pipe = EMOAnimationPipeline(vae=vae, unet=unet, scheduler=DDIMScheduler())
    got_lat = pipe.images2latents(img[None])
    assert got_lat.shape == (1, 4, 16, 16)
    torch.testing.assert_close(got_lat.cpu(), want_lat, rtol=1e-3, atol=1e-4)
    kw = dict(height=128, width=128, num_inference_steps=2, appearance_encoder=ref, context_frames=4, init_latents=seeded_randn((4, 4, 16, 16), 5),
              text_embeddings=seeded_randn((2, 5, 32), 2), output_type="latent")
    a = pipe("", 4, source_image=img, **kw).videos
    b = pipe("", 4, ref_image_latents=want_lat, **kw).videos
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=2e-3, atol=2e-4)
    path = str(tmp_path / "ref.png")
    Image.fromarray(rng.randint(0, 256, size=(96, 160, 3)).astype(np.uint8)).save(path)     # resized to (width, height) by the path branch
    c = pipe("", 4, source_image=path, **kw).videos
    assert tuple(c.shape) == (1, 4, 4, 16, 16) and bool(torch.isfinite(c).all())
    with pytest.raises(ValueError):
        pipe("", 4, source_image=img[:, :, :2], **kw)
