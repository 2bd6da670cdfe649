"""Oracle (test infrastructure): functional fp32 CPU restatement of the reference's VideoNet path (/root/reference/models/videonet.py).

  spatial_attention_module   SpatialAttentionModule.forward            :39-77
  rcab                       ReferenceConditionedAttentionBlock.forward :165-196  (sam -> cross_attn -> tam)
  videonet_forward           VideoNet.forward / update_reference_embeddings :236-267

Third-party members of the reference classes and how they are restated: `xformers.ops.memory_efficient_attention` by its published
semantics softmax(q k^T d^-0.5) v; `cross_attn` (a diffusers Transformer2DModel) and the UNet2DConditionModel around it by the
one-frame instance of the pinned 3-D blocks (oracle/unet_ref.py: Transformer3D at F = 1 == Transformer2D, ResnetBlock3D at F = 1 ==
the 2-D resnet - parity unpinned against diffusers by construction, like SURVEY A15); `get_motion_module` (models/motionmodule.py,
diffusers Attention / FeedForward inside) by oracle/unet_ref.motion_module - the same module tree as
magicanimate/models/motion_module.py - with the ctor defaults of models/motionmodule.py:37-48 (8 heads, TWO transformer blocks,
no positional encoding).

PINNED by tests/golden/videonet.safetensors (`spatial/out`, `rcab/*`: the reference's own class bodies, AST-extracted,
tools/oracle/gen_golden.py gen_videonet) and tests/golden/videonet_wiring.json (block order, dealing of the reference embeddings,
state-dict key listing from the reference's VideoNet ctor)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import unet_ref as U

TAM_HEADS = 8      # models/motionmodule.py:40 num_attention_heads default (VideoNet passes motion_module_kwargs={}, videonet.py:151-153)
SAM_HEADS = 8      # models/videonet.py:16 num_heads default


def spatial_attention_module(sd, p, x, ref, heads=SAM_HEADS):
    """models/videonet.py:39-77.  x, ref (bt, C, h, w)."""
    bt, C, h, w = x.shape
    cat = torch.cat((x, ref), dim=3)                                                          # :42 concat along the WIDTH
    proj = F.conv2d(F.group_norm(cat, 32, sd[p + ".norm_in.weight"], sd[p + ".norm_in.bias"], 1e-6), sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    grouped = proj.permute(0, 2, 3, 1).reshape(bt, h * 2 * w, C)                              # :50
    xr = x.permute(0, 2, 3, 1).reshape(bt, h * w, C)                                          # :51
    lin = lambda n, t: F.linear(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"])
    q, k, v = lin("to_q", xr), lin("to_k", grouped), lin("to_v", grouped)                     # :54  q from the RAW tokens
    d = q.shape[-1] // heads
    sp = lambda t: t.reshape(bt, -1, heads, d).permute(0, 2, 1, 3)
    a = torch.matmul((torch.matmul(sp(q), sp(k).transpose(-1, -2)) * d ** -0.5).softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(bt, h * w, -1)
    n1 = F.layer_norm(a + xr, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])           # :66
    n2 = F.layer_norm(n1 + lin("ffn", n1), (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    out = n2.reshape(bt, h, w, C).permute(0, 3, 1, 2)
    return F.conv2d(out, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]) + x           # :74-77


def rcab(sd, p, x, ref, ctx, heads, groups, num_frames, skip_temporal_attn=False, use_linear_projection=False):
    """ReferenceConditionedAttentionBlock.forward (models/videonet.py:165-196).  x, ref (bt, C, h, w); ctx (bt, L, D)."""
    out = spatial_attention_module(sd, p + ".sam", x, ref)                                    # :181
    out = U.transformer3d(sd, p + ".cross_attn", out[:, :, None], ctx, heads, groups, use_linear_projection)[:, :, 0]   # :184-185
    if skip_temporal_attn:                                                                    # :188-189
        return out
    bt, C, h, w = out.shape
    t = out.reshape(bt // num_frames, num_frames, C, h, w).permute(0, 2, 1, 3, 4)             # :192 '(b t) c h w -> b c t h w'
    t = U.motion_module(sd, p + ".tam", t, TAM_HEADS)                                         # :195
    return t.permute(0, 2, 1, 3, 4).reshape(bt, C, h, w)                                      # :198


def block_order(cfg):
    """Order in which VideoNet.__init__ (models/videonet.py:216-234) collects its blocks = order in which
    update_reference_embeddings (:237-244) deals `reference_embeddings[i]`: down blocks, mid block, up blocks."""
    c = U.normalize_config(cfg)
    out = []
    for i, t in enumerate(c["down_block_types"]):
        if t.startswith("CrossAttn"):
            out += [f"down_blocks.{i}.attentions.{j}" for j in range(c["layers_per_block"])]
    out.append("mid_block.attentions.0")
    for i, t in enumerate(c["up_block_types"]):
        if t.startswith("CrossAttn"):
            out += [f"up_blocks.{i}.attentions.{j}" for j in range(c["layers_per_block"] + 1)]
    return out


def videonet_forward(sd, cfg, initial_noise, timesteps, reference_embeddings, clip_condition_embeddings, num_frames, skip_temporal_attn=False):
    """VideoNet.forward (models/videonet.py:252-267): deal the reference embeddings, set skip_temporal_attn, run the 2-D UNet on the
    (b*t, 4, h, w) batch.  `sd` carries the reference's keys (unet.<block>.attentions.j.{cross_attn,sam,tam}...)."""
    sd_u = {k[5:] if k.startswith("unet.") else k: v for k, v in sd.items()}
    c = U.normalize_config(cfg)
    refs = dict(zip(block_order(cfg), reference_embeddings))

    def hook(p, x, ctx, heads):
        y = rcab(sd_u, p, x[:, :, 0], refs[p], ctx, heads, c["norm_num_groups"], num_frames, skip_temporal_attn, c["use_linear_projection"])
        return y[:, :, None]
    return U.unet_forward(sd_u, cfg, initial_noise[:, :, None], timesteps, clip_condition_embeddings, attention_hook=hook)[:, :, 0]
