"""Oracle (test infrastructure): functional fp32 CPU restatement of the Backbone UNet forward.

State-dict driven: `unet_forward(sd, cfg, ...)` takes the reference's state-dict keys
(SURVEY.md section 8b "Weights") and a config dict with the reference ctor's kwargs
(magicanimate/models/unet_controlnet.py:57-105).  Layout is the reference's
(B, C, F, H, W); all arithmetic is torch.nn.functional on CPU.

Each function cites the reference file:line it restates.  Pinned by
tests/test_oracle_golden.py against tests/golden/*.safetensors.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- config

_DEFAULTS = dict(  # unet_controlnet.py:57-105
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False,
    flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    mid_block_type="UNetMidBlock3DCrossAttn",
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=1280, attention_head_dim=8, dual_cross_attention=False,
    use_linear_projection=False, class_embed_type=None, num_class_embeds=None, upcast_attention=False,
    resnet_time_scale_shift="default", use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=False, motion_module_decoder_only=False, motion_module_type=None,
    motion_module_kwargs={}, unet_use_cross_frame_attention=None, unet_use_temporal_attention=None,
)


def normalize_config(cfg):
    out = dict(_DEFAULTS)
    for k, v in cfg.items():
        if k in out:
            out[k] = v
    out["norm_eps"] = float(out["norm_eps"])  # PyYAML reads '1e-05' as str
    n = len(out["block_out_channels"])
    hd = out["attention_head_dim"]
    out["attention_head_dim"] = tuple(hd) if isinstance(hd, (list, tuple)) else (hd,) * n
    return out


# ----------------------------------------------------------------------------- primitives

def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """embeddings.py:28-68 (get_timestep_embedding); fp32 table."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    ang = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    s, c = torch.sin(ang), torch.cos(ang)
    emb = torch.cat([c, s], -1) if flip_sin_to_cos else torch.cat([s, c], -1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv_per_frame(sd, p, x, stride=1, padding=1):
    """resnet.py:30-38 (InflatedConv3d): 2-D conv applied to every frame."""
    b, c, f, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), sd[p + ".weight"], sd.get(p + ".bias"),
                 stride=stride, padding=padding)
    return y.reshape(b, f, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def group_norm_5d(sd, p, x, groups, eps):
    """resnet.py:180,191 / unet_controlnet.py:476: nn.GroupNorm on a 5-D tensor => statistics
    are JOINT over (C/G, F, H, W)."""
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet_block(sd, p, x, emb, groups, eps, scale=1.0):
    """resnet.py:177-207 (ResnetBlock3D.forward), time_embedding_norm='default'."""
    h = F.silu(group_norm_5d(sd, p + ".norm1", x, groups, eps))
    h = _conv_per_frame(sd, p + ".conv1", h)
    h = h + _lin(sd, p + ".time_emb_proj", F.silu(emb))[:, :, None, None, None]
    h = F.silu(group_norm_5d(sd, p + ".norm2", h, groups, eps))
    h = _conv_per_frame(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv_per_frame(sd, p + ".conv_shortcut", x, padding=0)
    return (x + h) / scale


MAX_SCORE_ELEMS = 1 << 30   # 4 GB of f32 scores per attention chunk


def attention(sd, p, x, ctx, heads, upcast=False):
    """orig_attention.py:598-684 (CrossAttention.forward/_attention): q/k/v no bias, scale d^-0.5,
    softmax over keys, to_out[0] Linear+bias."""
    ctx = x if ctx is None else ctx
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx)
    b, lq, c = q.shape
    d = c // heads
    sp = lambda t: t.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    if upcast:
        q, k = q.float(), k.float()
    # batch rows are independent: the score tensor is formed for MAX_SCORE_ELEMS elements' worth of rows at a time (the full-size
    # level-0 read pass would otherwise hold 12 x 8 x 4096 x 8192 f32 scores = 12.9 GB twice) - the same per-row arithmetic
    step = max(1, MAX_SCORE_ELEMS // max(1, heads * lq * k.shape[2]))
    o = torch.cat([torch.matmul((torch.matmul(q[i:i + step], k[i:i + step].transpose(-1, -2)) * (d ** -0.5)).softmax(-1).to(v.dtype),
                                v[i:i + step]) for i in range(0, b, step)], 0)
    o = o.permute(0, 2, 1, 3).reshape(b, lq, c)
    return _lin(sd, p + ".to_out.0", o)


def feed_forward(sd, p, x):
    """orig_attention.py:776-781,817-827: GEGLU (value, gate) chunk order, exact-erf GELU, Linear."""
    h, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", h * F.gelu(gate))


class GuttedBlock(Exception):
    """Raised by the ReferenceNet's last transformer block (appearance_encoder.py:613-621: attn1 projections parameter-less,
    attn2 None, norm2 / norm3 / ff / proj_out Identity): behind its LN1 - the last bank - nothing carries parameters and the
    model output is discarded (EMOAnimationPipeline.py:711-716), so the oracle's write pass ends there."""


def basic_transformer_block(sd, p, x, ctx, heads, frames, bank=None, bank_mode=None, uc_rows=None,
                            written=None, upcast=False):
    """attention.py:276-320 (plain) and mutual_self_attention.py:199-284 (write / read hooks).

    x: (B*F, L, C).  bank_mode None|'write'|'read'.  In 'write' the LN1 output is appended to
    `written`.  In 'read', `bank` is (B_ref, L_ref, C): repeated over F, truncated to B*F rows
    (mutual_self_attention.py:238) and concatenated to the keys/values (:239-241); rows listed
    in `uc_rows` (bool mask over B*F) are recomputed with plain self-attention (:243-256)."""
    n1 = F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    if bank_mode == "write":
        written.append(n1.clone())
    if (p + ".attn1.to_q.weight") not in sd:
        raise GuttedBlock(p)
    if bank_mode == "read" and bank is not None:
        rep = bank.unsqueeze(1).repeat(1, frames, 1, 1).reshape(-1, *bank.shape[1:])[: x.shape[0]]
        h = attention(sd, p + ".attn1", n1, torch.cat([n1, rep], dim=1), heads, upcast) + x
        if uc_rows is not None and bool(uc_rows.any()):
            h = h.clone()
            h[uc_rows] = attention(sd, p + ".attn1", n1[uc_rows], None, heads, upcast) + x[uc_rows]
        x = h
    else:
        x = attention(sd, p + ".attn1", n1, None, heads, upcast) + x
    if (p + ".attn2.to_q.weight") in sd:
        n2 = F.layer_norm(x, x.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
        x = attention(sd, p + ".attn2", n2, ctx, heads, upcast) + x
    n3 = F.layer_norm(x, x.shape[-1:], sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    return feed_forward(sd, p + ".ff", n3) + x


def transformer3d(sd, p, x, ctx, heads, groups, use_linear_projection=False, **kw):
    """attention.py:112-161 (Transformer3DModel.forward): per-frame GroupNorm eps 1e-6,
    proj_in (1x1 conv | Linear), one BasicTransformerBlock, proj_out, + residual."""
    b, c, f, hh, ww = x.shape
    xf = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, hh, ww)
    if ctx.shape[0] != xf.shape[0]:  # attention.py:118-119
        ctx = ctx.repeat_interleave(f, dim=0)
    h = F.group_norm(xf, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(b * f, hh * ww, c)
    w_in = sd[p + ".proj_in.weight"].reshape(c, c)  # conv 1x1 == linear over channels
    h = F.linear(h, w_in, sd[p + ".proj_in.bias"])
    h = basic_transformer_block(sd, p + ".transformer_blocks.0", h, ctx, heads, f, **kw)
    h = F.linear(h, sd[p + ".proj_out.weight"].reshape(c, c), sd[p + ".proj_out.bias"])
    h = h.reshape(b * f, hh, ww, c).permute(0, 3, 1, 2) + xf
    return h.reshape(b, f, c, hh, ww).permute(0, 2, 1, 3, 4)


def positional_encoding(d_model, max_len=24):
    """motion_module.py:237-245: pe[0,:,0::2]=sin, pe[0,:,1::2]=cos."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def motion_module(sd, p, x, heads, groups=32, n_attn=2):
    """motion_module.py:82-87,139-163,215-227,275-334 (VanillaTemporalModule): per-frame GN(32,
    eps 1e-6) -> Linear -> num_transformer_block x { [LN -> (tokens (b hw) f c, +PE, temporal self-attn) -> +res] x n_attn
    -> LN -> GEGLU FF -> +res } -> Linear -> + residual.  The block count is read off the state dict
    (configs/inference.yaml: 1; the ctor default, which models/videonet.py:151-153 takes: 2)."""
    p = p + ".temporal_transformer"
    b, c, f, hh, ww = x.shape
    xf = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, hh, ww)
    h = F.group_norm(xf, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(b * f, hh * ww, c)
    h = _lin(sd, p + ".proj_in", h)
    d = hh * ww
    bi = 0
    while f"{p}.transformer_blocks.{bi}.ff_norm.weight" in sd:
        tb = f"{p}.transformer_blocks.{bi}"
        for k in range(n_attn):
            n = F.layer_norm(h, (c,), sd[f"{tb}.norms.{k}.weight"], sd[f"{tb}.norms.{k}.bias"])
            t = n.reshape(b, f, d, c).permute(0, 2, 1, 3).reshape(b * d, f, c)   # (b f) d c -> (b d) f c
            pe_key = f"{tb}.attention_blocks.{k}.pos_encoder.pe"
            if pe_key in sd:
                t = t + sd[pe_key][:, :f]
            a = attention(sd, f"{tb}.attention_blocks.{k}", t, None, heads)
            a = a.reshape(b, d, f, c).permute(0, 2, 1, 3).reshape(b * f, d, c)
            h = a + h
        n = F.layer_norm(h, (c,), sd[tb + ".ff_norm.weight"], sd[tb + ".ff_norm.bias"])
        h = feed_forward(sd, tb + ".ff", n) + h
        bi += 1
    h = _lin(sd, p + ".proj_out", h)
    h = h.reshape(b * f, hh, ww, c).permute(0, 3, 1, 2) + xf
    return h.reshape(b, f, c, hh, ww).permute(0, 2, 1, 3, 4)


def upsample(sd, p, x, output_size=None):
    """resnet.py:56-84 (Upsample3D): nearest x(1,2,2) - or to an explicit `output_size` (f, h, w) when the UNet forwards the
    skip's size (unet_controlnet.py:357-365,456-459) - then conv3x3."""
    if output_size is None:
        x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
    else:
        x = F.interpolate(x, size=tuple(output_size), mode="nearest")
    return _conv_per_frame(sd, p + ".conv", x)


# ----------------------------------------------------------------------------- bank bookkeeping

def transformer_block_order(cfg, fusion_blocks="midup"):
    """Prefixes of the BasicTransformerBlocks that take part in reference read/write, in the
    reference's pairing order: torch_dfs(mid)+torch_dfs(up) [or torch_dfs(unet)] stable-sorted by
    descending width (mutual_self_attention.py:532-543,585-586)."""
    cfg = normalize_config(cfg)
    boc = cfg["block_out_channels"]
    items = []
    if fusion_blocks == "full":
        for i, t in enumerate(cfg["down_block_types"]):
            if t.startswith("CrossAttn"):
                for j in range(cfg["layers_per_block"]):
                    items.append((f"down_blocks.{i}.attentions.{j}", boc[i]))
    # module registration order of the reference ctor is down_blocks, up_blocks, mid_block
    # (`self.mid_block = None` first, the module is assigned after both ModuleLists exist -
    # unet_controlnet.py:136-138,182), so torch_dfs(unet) visits mid LAST; 'midup' lists mid first.
    if fusion_blocks == "midup":
        items.append(("mid_block.attentions.0", boc[-1]))
    rev = list(reversed(boc))
    for i, t in enumerate(cfg["up_block_types"]):
        if t.startswith("CrossAttn"):
            for j in range(cfg["layers_per_block"] + 1):
                items.append((f"up_blocks.{i}.attentions.{j}", rev[i]))
    if fusion_blocks == "full":
        items.append(("mid_block.attentions.0", boc[-1]))
    items = sorted(items, key=lambda it: -it[1])  # python sort is stable
    return [n for n, _ in items]


# ----------------------------------------------------------------------------- the UNet

def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, *, bank_mode=None, banks=None,
                 uc_rows=None, fusion_blocks="midup", down_block_additional_residuals=None,
                 mid_block_additional_residual=None, speed_embeddings=None, audio_features=None,
                 return_banks=False, attention_hook=None, class_labels=None):
    """unet_controlnet.py:328-483 (UNet3DConditionModel.forward).

    bank_mode='write': returns (sample, {block_prefix: LN1 output}) - the ReferenceNet pass.
    bank_mode='read' : `banks` maps block_prefix -> (B_ref, L, C) tensor; `uc_rows` bool (B*F,).
    EMO extension (no reference behaviour): `audio_features` (B*F, L_a, D) replaces
    encoder_hidden_states as the per-frame attn2 context; `speed_embeddings` (B, 4*C0) is added to
    the time embedding (class-embedding slot, unet_controlnet.py:400-408).
    attention_hook(prefix, x, ctx, heads): replaces the transformer of every attention slot (models/videonet.py:216-234 puts a
    ReferenceConditionedAttentionBlock there - oracle/videonet_ref.py)."""
    cfg = normalize_config(cfg)
    boc, G, eps = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["norm_eps"]
    hd = cfg["attention_head_dim"]
    B, _, Fr, _, _ = sample.shape
    mmk = cfg["motion_module_kwargs"] or {}
    mm_heads = mmk.get("num_attention_heads", 8)
    mm_nattn = len(mmk.get("attention_block_types", ("Temporal_Self", "Temporal_Self")))
    ulp, upc = cfg["use_linear_projection"], cfg["upcast_attention"]
    if cfg["center_input_sample"]:
        sample = 2 * sample - 1.0
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
    elif timestep.dim() == 0:
        timestep = timestep[None]
    timestep = timestep.expand(B)
    t_emb = timestep_embedding(timestep, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    t_emb = t_emb.to(sd["time_embedding.linear_1.weight"].dtype)      # unet_controlnet.py:397 `t_emb.to(dtype=self.dtype)` (a no-op in f32)
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))
    if cfg["class_embed_type"] is not None or cfg["num_class_embeds"] is not None:      # unet_controlnet.py:400-408
        if class_labels is None:
            raise ValueError("class_labels should be provided when num_class_embeds > 0")
        if cfg["class_embed_type"] == "timestep":
            ce = timestep_embedding(class_labels.reshape(-1), boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(emb.dtype)
            ce = _lin(sd, "class_embedding.linear_2", F.silu(_lin(sd, "class_embedding.linear_1", ce)))
        elif cfg["class_embed_type"] == "identity":
            ce = class_labels
        else:
            ce = F.embedding(class_labels, sd["class_embedding.weight"])
        emb = emb + ce.to(emb.dtype)
    if speed_embeddings is not None:
        emb = emb + speed_embeddings
    ctx = encoder_hidden_states if audio_features is None else audio_features

    active = set(transformer_block_order(cfg, fusion_blocks)) if bank_mode else set()
    written = {}

    def tf(p, x, heads):
        if attention_hook is not None:
            return attention_hook(p, x, ctx, heads)
        kw = dict(upcast=upc)
        if p in active:
            if bank_mode == "write":
                lst = []
                kw.update(bank_mode="write", written=lst)
                try:
                    y = transformer3d(sd, p, x, ctx, heads, G, ulp, **kw)
                finally:
                    if lst:
                        written[p] = lst[0]
                return y
            kw.update(bank_mode="read", bank=(banks or {}).get(p), uc_rows=uc_rows)
        return transformer3d(sd, p, x, ctx, heads, G, ulp, **kw)

    def mm(p, x):
        return motion_module(sd, p, x, mm_heads, 32, mm_nattn) if (p + ".temporal_transformer.norm.weight") in sd else x

    x = _conv_per_frame(sd, "conv_in", sample)
    skips = [x]
    for i, t in enumerate(cfg["down_block_types"]):
        p = f"down_blocks.{i}"
        for j in range(cfg["layers_per_block"]):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, G, eps)
            if t.startswith("CrossAttn"):
                x = tf(f"{p}.attentions.{j}", x, hd[i])
            x = mm(f"{p}.motion_modules.{j}", x)
            skips.append(x)
        if (p + ".downsamplers.0.conv.weight") in sd:
            x = _conv_per_frame(sd, p + ".downsamplers.0.conv", x, stride=2, padding=cfg["downsample_padding"])
            skips.append(x)
    if down_block_additional_residuals is not None and mid_block_additional_residual is not None:
        skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]

    sc = cfg["mid_block_scale_factor"]
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, G, eps, sc)
    x = tf("mid_block.attentions.0", x, hd[-1])
    x = mm("mid_block.motion_modules.0", x)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, G, eps, sc)
    if down_block_additional_residuals is not None and mid_block_additional_residual is not None:
        x = x + mid_block_additional_residual

    rhd = list(reversed(hd))
    forward_upsample_size = any(s_ % (2 ** (len(boc) - 1)) != 0 for s_ in sample.shape[-2:])   # unet_controlnet.py:357-365
    try:
        for i, t in enumerate(cfg["up_block_types"]):
            p = f"up_blocks.{i}"
            for j in range(cfg["layers_per_block"] + 1):
                x = torch.cat([x, skips.pop()], dim=1)  # unet_3d_blocks.py:627-629,729-731
                x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, G, eps)
                if t.startswith("CrossAttn"):
                    x = tf(f"{p}.attentions.{j}", x, rhd[i])
                x = mm(f"{p}.motion_modules.{j}", x)
            if (p + ".upsamplers.0.conv.weight") in sd:
                x = upsample(sd, p + ".upsamplers.0", x, tuple(skips[-1].shape[2:]) if (forward_upsample_size and skips) else None)
    except GuttedBlock:
        if bank_mode != "write":
            raise
        return None, written

    if "conv_out.weight" in sd:  # the AppearanceEncoder has no conv_norm_out/conv_out
        x = F.silu(group_norm_5d(sd, "conv_norm_out", x, G, eps))
        x = _conv_per_frame(sd, "conv_out", x)
    if bank_mode == "write" or return_banks:
        return x, written
    return x


def round_banks_fp16(written):
    """mutual_self_attention.py:577,588: reader.bank = [v.clone().to(float16)] - banks are rounded
    through fp16 even in an fp32 run."""
    return {k: v.to(torch.float16).to(v.dtype) for k, v in written.items()}
