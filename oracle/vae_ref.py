"""Oracle (test infrastructure): AutoencoderKL of the SD-1.x family, restated in plain PyTorch fp32 on the CPU.

PARITY UNPINNED: the reference imports the VAE from third-party `diffusers` (not in tree, version unpinned -
requirements.txt:1; call sites EMOAnimationPipeline.py:291-307 decode_latents, :402-414 encode of the reference image).
Restated from the published architecture (Rombach et al. 2022, the `first_stage_model` of latent-diffusion / diffusers'
AutoencoderKL with block_out_channels (128, 256, 512, 512), layers_per_block 2, 32 GroupNorm groups, eps 1e-6) under
diffusers' state-dict key names.  Only tests/ import this module; the product path is emote_hack_amd/vae.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

GROUPS, EPS = 32, 1e-6


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], EPS)


def resnet(sd, p, x, groups=GROUPS):
    """ResnetBlock2D without time embedding: GN -> SiLU -> conv -> GN -> SiLU -> conv (+ 1x1 shortcut), scale 1"""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def mid_attention(sd, p, x, groups=GROUPS):
    """one head over the h*w tokens: GN -> q/k/v Linear (bias) -> softmax(q k^T / sqrt(C)) v -> out Linear -> + x"""
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".group_norm", x, groups).reshape(b, c, hh * ww).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * (c ** -0.5), dim=-1) @ v
    a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return a.transpose(1, 2).reshape(b, c, hh, ww) + x


def mid_block(sd, p, x, groups=GROUPS):
    x = resnet(sd, p + ".resnets.0", x, groups)
    x = mid_attention(sd, p + ".attentions.0", x, groups)
    return resnet(sd, p + ".resnets.1", x, groups)


def decode(sd, z, n_blocks=4, layers_per_block=2, groups=GROUPS):
    """AutoencoderKL.decode: z (n, 4, h, w) -> (n, 3, 8h, 8w)"""
    x = _conv(sd, "post_quant_conv", z, padding=0)
    x = _conv(sd, "decoder.conv_in", x)
    x = mid_block(sd, "decoder.mid_block", x, groups)
    for i in range(n_blocks):
        for j in range(layers_per_block + 1):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, groups)
        if (f"decoder.up_blocks.{i}.upsamplers.0.conv.weight") in sd:
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, groups))
    return _conv(sd, "decoder.conv_out", x)


def encode(sd, x, n_blocks=4, layers_per_block=2, groups=GROUPS):
    """AutoencoderKL.encode: x (n, 3, H, W) -> moments (n, 8, H/8, W/8) = [mean | logvar]"""
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(n_blocks):
        for j in range(layers_per_block):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, groups)
        if (f"encoder.down_blocks.{i}.downsamplers.0.conv.weight") in sd:
            h = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = mid_block(sd, "encoder.mid_block", h, groups)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, groups))
    h = _conv(sd, "encoder.conv_out", h)
    return _conv(sd, "quant_conv", h, padding=0)


def decode_latents(sd, latents, scaling_factor=0.18215, **kw):
    """EMOAnimationPipeline.py:291-307: latents (b, 4, f, h, w) -> video (b, 3, f, H, W) in [0, 1], one frame per decode call."""
    b, c, f, h, w = latents.shape
    lat = (latents / scaling_factor).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    video = torch.cat([decode(sd, lat[i:i + 1], **kw) for i in range(lat.shape[0])])
    video = video.reshape(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1)
