"""Shared definitions of the parity cases: configs, seeds and input constructors.

Imported by tools/oracle/gen_golden.py (build container; writes tests/golden/*) and by the tests
(CPU: oracle vs golden; GPU: HIP path vs golden and vs oracle).  Reference-free.
"""
from __future__ import annotations

import os

import torch

from emote_hack_amd.synth import seeded_randn

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")

MOTION_KW_TINY = dict(num_attention_heads=4, num_transformer_block=1,
                      attention_block_types=["Temporal_Self", "Temporal_Self"],
                      temporal_position_encoding=True, temporal_position_encoding_max_len=24,
                      temporal_attention_dim_div=1)

# probe-validated tiny config (Backbone 4.71 M params)
TINY = dict(sample_size=16, block_out_channels=(32, 64, 64, 64), norm_num_groups=8, attention_head_dim=4,
            cross_attention_dim=32, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
TINY_MOTION = dict(TINY, use_motion_module=True, motion_module_type="Vanilla", motion_module_kwargs=MOTION_KW_TINY)
TINY_LINEAR = dict(TINY_MOTION, use_linear_projection=True, upcast_attention=True, attention_head_dim=(2, 4, 4, 8))

MOTION_KW_FULL = dict(num_attention_heads=8, num_transformer_block=1,
                      attention_block_types=["Temporal_Self", "Temporal_Self"],
                      temporal_position_encoding=True, temporal_position_encoding_max_len=24,
                      temporal_attention_dim_div=1)  # configs/inference.yaml:11-21

# BASELINE config 2 backbone: SD-1.5 widths + motion modules (configs/unet-config.yaml:default with
# norm_num_groups=32 as shipped SD-1.5 checkpoints use; inference.yaml motion kwargs)
SD15 = dict(sample_size=64, block_out_channels=(320, 640, 1280, 1280), norm_num_groups=32, attention_head_dim=8,
            cross_attention_dim=768, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
SD15_MOTION = dict(SD15, use_motion_module=True, motion_module_type="Vanilla", motion_module_kwargs=MOTION_KW_FULL)

REF_PREFIX = "reference_unet."  # seed salt for ReferenceNet weights


def tiny_inputs(batch=2, frames=4, hw=16, ctx_len=5, ctx_dim=32):
    x = seeded_randn((batch, 4, frames, hw, hw), 1)
    ctx = seeded_randn((batch, ctx_len, ctx_dim), 2)
    return x, ctx


def uc_rows(batch, frames):
    m = torch.zeros(batch * frames, dtype=torch.bool)
    m[: (batch // 2) * frames] = True
    return m


WINDOW_CASES = [(12, 16, 1, 4), (24, 16, 1, 4), (48, 16, 1, 4), (48, 12, 1, 0), (24, 12, 1, 4),
                (48, 16, 2, 4), (40, 16, 3, 4), (17, 16, 1, 4)]  # (F_tot, ctx, stride, overlap)

SPEEDS = [-1.0, -0.13, 0.0, 0.07, 0.49, 2.0, 0.125, -0.125, 0.375, -0.875, -3.0, 0.999]

# a 2-layer member of the wav2vec2-base family (feat_extract_norm "group", post-LN): fast CPU check of oracle/wav2vec2_ref.py
WAV2VEC2_TINY = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, conv_dim=(32,) * 7,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)

# models/videonet.py: the 2-D UNet a VideoNet starts from (no motion modules; 8-head sam / tam need channels that are multiples of 64)
VIDEONET_TINY = dict(sample_size=16, block_out_channels=(64, 64, 128, 128), norm_num_groups=32, attention_head_dim=8, cross_attention_dim=32,
                     unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
