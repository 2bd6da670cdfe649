"""Oracle (test infrastructure): the small EMO conditioning modules,
functional over state dicts.  Pinned by tests/golden/conditioning.safetensors (captured by
AST-extracting the class bodies from the reference files whose module imports fail here).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SPEED_ENCODER_CENTERS = [-1.0, -0.5, -0.2, -0.1, 0.0, 0.1, 0.2, 0.5, 1.0]  # Net.py:221-224
SPEED_ENCODER_RADIUS = 0.1                                                  # Net.py:226-229


def speed_encoder_encode(v):
    """Net.py:231-247 (SpeedEncoder.encode_speed): tanh((v - c)/r * 3) per bucket."""
    c = torch.tensor(SPEED_ENCODER_CENTERS, dtype=torch.float32)
    return torch.tanh((v[:, None] - c[None, :]) / SPEED_ENCODER_RADIUS * 3)


def speed_encoder(sd, v, p="mlp"):
    """Net.py:249-258: Linear(9->D) -> ReLU -> Linear(D->D)."""
    h = F.relu(F.linear(speed_encoder_encode(v), sd[f"{p}.0.weight"], sd[f"{p}.0.bias"]))
    return F.linear(h, sd[f"{p}.2.weight"], sd[f"{p}.2.bias"])


def speed_bucket_index(v, num_buckets=9):
    """train_stage_3_speedlayers.py:42-47 (SpeedController.map_speed_to_bucket): INT, bit-exact.
    argmin |v - linspace(-1,1,9)|; ties -> lower index; out of range clamps to the end buckets."""
    centers = torch.linspace(-1.0, 1.0, num_buckets)
    return torch.argmin(torch.abs(v.unsqueeze(-1) - centers), dim=-1)


def speed_controller(sd, v):
    """train_stage_3_speedlayers.py:49-55: Embedding(9,D)[bucket] -> Linear -> ReLU -> Linear."""
    e = sd["speed_embedding.weight"][speed_bucket_index(v, sd["speed_embedding.weight"].shape[0])]
    h = F.relu(F.linear(e, sd["speed_mlp.0.weight"], sd["speed_mlp.0.bias"]))
    return F.linear(h, sd["speed_mlp.2.weight"], sd["speed_mlp.2.bias"])


def face_region_controller(sd, mask):
    """train_stage_3_speedlayers.py:57-76: 4x conv3x3 (+ReLU between), 1->64->128->256->D."""
    h = mask
    for i in (0, 2, 4, 6):
        h = F.conv2d(h, sd[f"encoder.{i}.weight"], sd[f"encoder.{i}.bias"], padding=1)
        if i != 6:
            h = F.relu(h)
    return h


def net_cross_attention_layer(sd, latent, audio, p=""):
    """Net.py:263-303 (CrossAttentionLayer): single head, q/k/v Linear WITH bias,
    scores / sqrt(feature_dim), softmax, @ value.  (No skip here; AudioAttentionLayers adds it.)"""
    q = F.linear(latent, sd[p + "query.weight"], sd[p + "query.bias"])
    k = F.linear(audio, sd[p + "key.weight"], sd[p + "key.bias"])
    v = F.linear(audio, sd[p + "value.weight"], sd[p + "value.bias"])
    s = torch.matmul(q, k.transpose(-2, -1)) / (q.shape[-1] ** 0.5)
    return torch.matmul(F.softmax(s, dim=-1), v)


def net_audio_attention_layers(sd, latent, audio, num_layers):
    """Net.py:305-325: latent = layer(latent, audio) + latent, per layer."""
    for i in range(num_layers):
        latent = net_cross_attention_layer(sd, latent, audio, f"layers.{i}.") + latent
    return latent


def net_reference_attention_layer(sd, latent, ref):
    """Net.py:333-365: same single-head attention, residual inside."""
    return latent + net_cross_attention_layer(sd, latent, ref)


def stage2_audio_attention(sd, frames, audio, heads=8):
    """train_stage_2_temporal_audio.py:146-177 (AudioAttention): q=frame_proj(x), k=v=audio_proj(a)
    (shared), 8 heads, scale (C/heads)^-0.5, out_proj."""
    B, T, C = frames.shape
    q = F.linear(frames, sd["frame_proj.weight"], sd["frame_proj.bias"])
    kv = F.linear(audio, sd["audio_proj.weight"], sd["audio_proj.bias"])
    d = C // heads
    q = q.reshape(B, T, heads, d).permute(0, 2, 1, 3)
    kv = kv.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    a = (torch.matmul(q, kv.transpose(-2, -1)) * d ** -0.5).softmax(-1)
    x = torch.matmul(a, kv).transpose(1, 2).reshape(B, T, C)
    return F.linear(x, sd["out_proj.weight"], sd["out_proj.bias"])


def stage2_temporal_attention(sd, x, heads=8):
    """train_stage_2_temporal_audio.py:123-144 (TemporalAttention): fused qkv (no bias), scores
    multiplied by a learnable per-head `temperature` (init 1.0, NOT d^-0.5), proj."""
    B, T, C = x.shape
    qkv = F.linear(x, sd["qkv.weight"]).reshape(B, T, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (torch.matmul(q, k.transpose(-2, -1)) * sd["temperature"]).softmax(-1)
    y = torch.matmul(a, v).transpose(1, 2).reshape(B, T, C)
    return F.linear(y, sd["proj.weight"], sd["proj.bias"])


def stage3_combine(latents, face_feat, unet_out_fn, speed_embed):
    """train_stage_3_speedlayers.py:242-271 (EMOStage3.forward) combine rule:
    unet(latents + face_feat) + speed_embed[..., None, None]."""
    return unet_out_fn(latents + face_feat) + speed_embed.unsqueeze(-1).unsqueeze(-1)


# ----------------------------------------------------------------------------- Net.py placeholders
def net_reference_attention(sd, x, ref):
    """Net.py:1487-1511 (ReferenceAttention): q = Wq LN(x), k = Wk LN(ref), v = Wv ref, one head, scale channels^-0.5, no residual."""
    B, C, H, W = x.shape
    xf, rf = x.flatten(2).permute(0, 2, 1), ref.flatten(2).permute(0, 2, 1)
    ln = lambda t: F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"])
    q = F.linear(ln(xf), sd["q_proj.weight"], sd["q_proj.bias"])
    k = F.linear(ln(rf), sd["k_proj.weight"], sd["k_proj.bias"])
    v = F.linear(rf, sd["v_proj.weight"], sd["v_proj.bias"])
    out = F.softmax(q @ k.transpose(-2, -1) * C ** -0.5, dim=-1) @ v
    return out.permute(0, 2, 1).reshape(B, C, H, W)


def net_motion_module(sd, x, heads=8):
    """Net.py:1449-1485 (MotionModule + TemporalAttention) on (B, C, T, 1, 1): Conv3d over time, LayerNorm(channels), multi-head
    attention over T (nn.MultiheadAttention: biased in_proj, q scaled by d^-0.5, out_proj), + identity."""
    B, C, T, _, _ = x.shape
    k = sd["temporal_conv.weight"].shape[2]
    y = F.conv3d(x, sd["temporal_conv.weight"], sd["temporal_conv.bias"], padding=(k // 2, 0, 0))
    t = y.reshape(B, C, T).permute(0, 2, 1)                                           # (B, T, C)
    n = F.layer_norm(t, (C,), sd["temporal_attention.norm.weight"], sd["temporal_attention.norm.bias"])
    qkv = F.linear(n, sd["temporal_attention.attention.in_proj_weight"], sd["temporal_attention.attention.in_proj_bias"])
    d = C // heads
    sp = lambda u: u.reshape(B, T, heads, d).permute(0, 2, 1, 3)
    q, kk, v = (sp(u) for u in qkv.chunk(3, dim=-1))
    a = (F.softmax(q @ kk.transpose(-1, -2) * d ** -0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    a = F.linear(a, sd["temporal_attention.attention.out_proj.weight"], sd["temporal_attention.attention.out_proj.bias"])
    return a.permute(0, 2, 1).reshape(B, C, T, 1, 1) + x
