"""Structure walk of the Backbone UNet: which blocks exist for a config, and the state-dict
keys / shapes they own.  The key names are the drop-in contract (SURVEY.md section 8b "Weights");
tests/test_spec.py checks them against the key list captured from the reference model.

Reference wiring: magicanimate/models/unet_controlnet.py:107-257 (ctor), unet_3d_blocks.py
(block ctors), resnet.py:113-175, attention.py:48-110,164-248, motion_module.py:53-213.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional

from .config import normalize_unet_config


@dataclass
class ResnetSpec:
    prefix: str
    cin: int
    cout: int
    temb: int

    @property
    def has_shortcut(self):
        return self.cin != self.cout  # resnet.py:171-175


@dataclass
class TransformerSpec:
    prefix: str
    channels: int
    heads: int
    ctx_dim: int
    linear_proj: bool
    # ReferenceNet only (magicanimate/models/appearance_encoder.py:613-621): the LAST transformer of the up path keeps norm,
    # proj_in and norm1 (whose output is the last bank) and nothing else - attn1 projections are parameter-less, attn2 is None,
    # norm2 / norm3 / ff / proj_out are Identity.  Everything behind its LN1 is dead (the ReferenceNet output is discarded).
    gutted: bool = False
    # VideoNet only (models/videonet.py:132-196): the attention slot holds a ReferenceConditionedAttentionBlock - this transformer
    # is its `cross_attn` child (prefix ends in ".cross_attn"), with a SpatialAttentionModule `sam` in front and a motion module
    # `tam` behind it (prefixes of the siblings; None in a plain UNet)
    sam: Optional[str] = None
    tam: Optional["MotionSpec"] = None


@dataclass
class MotionSpec:
    prefix: str
    channels: int
    heads: int
    n_attn: int
    pe_len: Optional[int]
    n_blocks: int = 1      # TemporalTransformer3DModel num_layers (configs/inference.yaml: 1; models/motionmodule.py default: 2)


@dataclass
class BlockSpec:
    kind: str  # 'down' | 'mid' | 'up'
    prefix: str
    resnets: List[ResnetSpec] = field(default_factory=list)
    attentions: List[Optional[TransformerSpec]] = field(default_factory=list)
    motions: List[Optional[MotionSpec]] = field(default_factory=list)
    sampler: Optional[str] = None  # prefix of downsamplers.0 / upsamplers.0
    channels: int = 0


@dataclass
class UNetSpec:
    cfg: dict
    down: List[BlockSpec]
    mid: BlockSpec
    up: List[BlockSpec]
    has_out: bool = True
    controlnet: Optional[tuple] = None   # ControlNet: conditioning-embedding channels (16, 32, 96, 256); no up path


def build_spec(cfg_kwargs, *, has_out=True, controlnet=None, gut_last_transformer=False) -> UNetSpec:
    cfg = normalize_unet_config(dict(cfg_kwargs), strict=False)
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    hd = cfg["attention_head_dim"]
    mmk = cfg["motion_module_kwargs"]
    n = len(boc)

    def motion(prefix, c, enabled):
        if not enabled:
            return None
        heads = mmk["num_attention_heads"]
        head_dim = c // heads // mmk["temporal_attention_dim_div"]
        if heads * head_dim != c:
            raise NotImplementedError("motion module inner_dim != in_channels")
        return MotionSpec(prefix, c, heads, len(mmk["attention_block_types"]),
                          mmk["temporal_position_encoding_max_len"] if mmk["temporal_position_encoding"] else None,
                          int(mmk["num_transformer_block"]))

    def tfm(prefix, c, heads):
        if c % heads:
            raise ValueError("channels not divisible by heads")
        return TransformerSpec(prefix, c, heads, cfg["cross_attention_dim"], cfg["use_linear_projection"])

    down = []
    out_c = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        res = 2 ** i
        use_mm = cfg["use_motion_module"] and (res in cfg["motion_module_resolutions"]) and not cfg["motion_module_decoder_only"]
        b = BlockSpec("down", f"down_blocks.{i}", channels=out_c)
        for j in range(cfg["layers_per_block"]):
            b.resnets.append(ResnetSpec(f"{b.prefix}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb))
            b.attentions.append(tfm(f"{b.prefix}.attentions.{j}", out_c, hd[i]) if t.endswith("CrossAttnDownBlock3D") else None)
            b.motions.append(motion(f"{b.prefix}.motion_modules.{j}", out_c, use_mm))
        if i != n - 1:
            b.sampler = f"{b.prefix}.downsamplers.0"
        down.append(b)

    c = boc[-1]
    mid = BlockSpec("mid", "mid_block", channels=c)
    mid.resnets = [ResnetSpec("mid_block.resnets.0", c, c, temb), ResnetSpec("mid_block.resnets.1", c, c, temb)]
    mid.attentions = [tfm("mid_block.attentions.0", c, hd[-1])]
    mid.motions = [motion("mid_block.motion_modules.0", c, cfg["use_motion_module"] and cfg["motion_module_mid_block"])]

    if controlnet is not None:   # magicanimate/models/controlnet.py:155-262: down blocks + mid block only
        return UNetSpec(cfg, down, mid, [], False, tuple(controlnet))
    up = []
    rboc = list(reversed(boc))
    rhd = list(reversed(hd))
    out_c = rboc[0]
    for i, t in enumerate(cfg["up_block_types"]):
        res = 2 ** (3 - i)
        prev_c, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, n - 1)]
        use_mm = cfg["use_motion_module"] and (res in cfg["motion_module_resolutions"])
        b = BlockSpec("up", f"up_blocks.{i}", channels=out_c)
        L = cfg["layers_per_block"] + 1
        for j in range(L):
            skip_c = in_c if j == L - 1 else out_c
            r_in = prev_c if j == 0 else out_c
            b.resnets.append(ResnetSpec(f"{b.prefix}.resnets.{j}", r_in + skip_c, out_c, temb))
            b.attentions.append(tfm(f"{b.prefix}.attentions.{j}", out_c, rhd[i]) if t.endswith("CrossAttnUpBlock3D") else None)
            b.motions.append(motion(f"{b.prefix}.motion_modules.{j}", out_c, use_mm))
        if i != n - 1:
            b.sampler = f"{b.prefix}.upsamplers.0"
        up.append(b)
    if gut_last_transformer:   # appearance_encoder.py:613-621 names up_blocks[3].attentions[2]: the last transformer of the up path
        last = [a for b in up for a in b.attentions if a is not None]
        if last:
            last[-1].gutted = True
    return UNetSpec(cfg, down, mid, up, has_out)


def _resnet_shapes(r: ResnetSpec, d):
    p = r.prefix
    d[f"{p}.norm1.weight"] = (r.cin,)
    d[f"{p}.norm1.bias"] = (r.cin,)
    d[f"{p}.conv1.weight"] = (r.cout, r.cin, 3, 3)
    d[f"{p}.conv1.bias"] = (r.cout,)
    d[f"{p}.time_emb_proj.weight"] = (r.cout, r.temb)
    d[f"{p}.time_emb_proj.bias"] = (r.cout,)
    d[f"{p}.norm2.weight"] = (r.cout,)
    d[f"{p}.norm2.bias"] = (r.cout,)
    d[f"{p}.conv2.weight"] = (r.cout, r.cout, 3, 3)
    d[f"{p}.conv2.bias"] = (r.cout,)
    if r.has_shortcut:
        d[f"{p}.conv_shortcut.weight"] = (r.cout, r.cin, 1, 1)
        d[f"{p}.conv_shortcut.bias"] = (r.cout,)


def _ff_shapes(p, c, d):
    d[f"{p}.net.0.proj.weight"] = (8 * c, c)
    d[f"{p}.net.0.proj.bias"] = (8 * c,)
    d[f"{p}.net.2.weight"] = (c, 4 * c)
    d[f"{p}.net.2.bias"] = (c,)


def _attn_shapes(p, c, kv, d):
    d[f"{p}.to_q.weight"] = (c, c)
    d[f"{p}.to_k.weight"] = (c, kv)
    d[f"{p}.to_v.weight"] = (c, kv)
    d[f"{p}.to_out.0.weight"] = (c, c)
    d[f"{p}.to_out.0.bias"] = (c,)


def _sam_shapes(p, c, d):
    """SpatialAttentionModule(num_inp_channels=c, embed_dim=c) (models/videonet.py:15-37), module registration order."""
    d[f"{p}.norm_in.weight"] = (c,)
    d[f"{p}.norm_in.bias"] = (c,)
    d[f"{p}.proj_in.weight"] = (c, c, 1, 1)
    d[f"{p}.proj_in.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v"):
        d[f"{p}.{n}.weight"] = (c, c)
        d[f"{p}.{n}.bias"] = (c,)
    d[f"{p}.norm1.weight"] = (c,)
    d[f"{p}.norm1.bias"] = (c,)
    d[f"{p}.ffn.weight"] = (c, c)
    d[f"{p}.ffn.bias"] = (c,)
    d[f"{p}.norm2.weight"] = (c,)
    d[f"{p}.norm2.bias"] = (c,)
    d[f"{p}.proj_out.weight"] = (c, c, 1, 1)
    d[f"{p}.proj_out.bias"] = (c,)


def _transformer_shapes(t: TransformerSpec, d):
    p, c = t.prefix, t.channels
    d[f"{p}.norm.weight"] = (c,)
    d[f"{p}.norm.bias"] = (c,)
    d[f"{p}.proj_in.weight"] = (c, c) if t.linear_proj else (c, c, 1, 1)
    d[f"{p}.proj_in.bias"] = (c,)
    tb = f"{p}.transformer_blocks.0"
    if t.gutted:   # key order of the reference module tree: attn1 (no parameters left), norm1
        d[f"{tb}.norm1.weight"] = (c,)
        d[f"{tb}.norm1.bias"] = (c,)
        return
    _attn_shapes(f"{tb}.attn1", c, c, d)
    d[f"{tb}.norm1.weight"] = (c,)
    d[f"{tb}.norm1.bias"] = (c,)
    _attn_shapes(f"{tb}.attn2", c, t.ctx_dim, d)
    d[f"{tb}.norm2.weight"] = (c,)
    d[f"{tb}.norm2.bias"] = (c,)
    _ff_shapes(f"{tb}.ff", c, d)
    d[f"{tb}.norm3.weight"] = (c,)
    d[f"{tb}.norm3.bias"] = (c,)
    d[f"{p}.proj_out.weight"] = (c, c) if t.linear_proj else (c, c, 1, 1)
    d[f"{p}.proj_out.bias"] = (c,)


def _motion_shapes(m: MotionSpec, d):
    p, c = m.prefix + ".temporal_transformer", m.channels
    d[f"{p}.norm.weight"] = (c,)
    d[f"{p}.norm.bias"] = (c,)
    d[f"{p}.proj_in.weight"] = (c, c)
    d[f"{p}.proj_in.bias"] = (c,)
    for b in range(m.n_blocks):
        tb = f"{p}.transformer_blocks.{b}"
        for k in range(m.n_attn):
            _attn_shapes(f"{tb}.attention_blocks.{k}", c, c, d)
            if m.pe_len:
                d[f"{tb}.attention_blocks.{k}.pos_encoder.pe"] = (1, m.pe_len, c)
            d[f"{tb}.norms.{k}.weight"] = (c,)
            d[f"{tb}.norms.{k}.bias"] = (c,)
        _ff_shapes(f"{tb}.ff", c, d)
        d[f"{tb}.ff_norm.weight"] = (c,)
        d[f"{tb}.ff_norm.bias"] = (c,)
    d[f"{p}.proj_out.weight"] = (c, c)
    d[f"{p}.proj_out.bias"] = (c,)


def skip_channels(spec: UNetSpec):
    """Channel counts of the skip tensors in push order: conv_in, every resnet sub-block, every downsampler."""
    out = [spec.cfg["block_out_channels"][0]]
    for b in spec.down:
        out += [b.channels] * len(b.resnets)
        if b.sampler:
            out.append(b.channels)
    return out


def param_shapes(spec: UNetSpec) -> "OrderedDict[str, tuple]":
    cfg = spec.cfg
    boc = cfg["block_out_channels"]
    d = OrderedDict()
    d["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    d["conv_in.bias"] = (boc[0],)
    d["time_embedding.linear_1.weight"] = (boc[0] * 4, boc[0])
    d["time_embedding.linear_1.bias"] = (boc[0] * 4,)
    d["time_embedding.linear_2.weight"] = (boc[0] * 4, boc[0] * 4)
    d["time_embedding.linear_2.bias"] = (boc[0] * 4,)
    # class embedding (unet_controlnet.py:119-127): nn.Embedding table | TimestepEmbedding | Identity (no parameters)
    if cfg.get("class_embed_type") is None and cfg.get("num_class_embeds") is not None:
        d["class_embedding.weight"] = (int(cfg["num_class_embeds"]), boc[0] * 4)
    elif cfg.get("class_embed_type") == "timestep":
        d["class_embedding.linear_1.weight"] = (boc[0] * 4, boc[0])
        d["class_embedding.linear_1.bias"] = (boc[0] * 4,)
        d["class_embedding.linear_2.weight"] = (boc[0] * 4, boc[0] * 4)
        d["class_embedding.linear_2.bias"] = (boc[0] * 4,)
    if spec.controlnet is not None:   # ControlNetConditioningEmbedding (controlnet.py:49-91), module registration order
        cc = spec.controlnet
        d["controlnet_cond_embedding.conv_in.weight"] = (cc[0], cfg.get("conditioning_channels", 3), 3, 3)
        d["controlnet_cond_embedding.conv_in.bias"] = (cc[0],)
        for i in range(len(cc) - 1):
            d[f"controlnet_cond_embedding.blocks.{2 * i}.weight"] = (cc[i], cc[i], 3, 3)
            d[f"controlnet_cond_embedding.blocks.{2 * i}.bias"] = (cc[i],)
            d[f"controlnet_cond_embedding.blocks.{2 * i + 1}.weight"] = (cc[i + 1], cc[i], 3, 3)
            d[f"controlnet_cond_embedding.blocks.{2 * i + 1}.bias"] = (cc[i + 1],)
        d["controlnet_cond_embedding.conv_out.weight"] = (boc[0], cc[-1], 3, 3)
        d["controlnet_cond_embedding.conv_out.bias"] = (boc[0],)
    for b in spec.down + ([] if spec.controlnet is not None else [spec.mid]) + spec.up:
        for r in b.resnets:
            _resnet_shapes(r, d)
        for a in b.attentions:
            if a is not None:
                _transformer_shapes(a, d)      # (VideoNet: ReferenceConditionedAttentionBlock registers cross_attn, sam, tam in this order)
                if a.sam is not None:
                    _sam_shapes(a.sam, a.channels, d)
                if a.tam is not None:
                    _motion_shapes(a.tam, d)
        for m in b.motions:
            if m is not None:
                _motion_shapes(m, d)
        if b.sampler:
            d[f"{b.sampler}.conv.weight"] = (b.channels, b.channels, 3, 3)
            d[f"{b.sampler}.conv.bias"] = (b.channels,)
    if spec.controlnet is not None:   # zero convs (controlnet.py:209-243), then the mid block (registered last, :245-257)
        for k, c in enumerate(skip_channels(spec)):
            d[f"controlnet_down_blocks.{k}.weight"] = (c, c, 1, 1)
            d[f"controlnet_down_blocks.{k}.bias"] = (c,)
        d["controlnet_mid_block.weight"] = (boc[-1], boc[-1], 1, 1)
        d["controlnet_mid_block.bias"] = (boc[-1],)
        for r in spec.mid.resnets:
            _resnet_shapes(r, d)
        for a in spec.mid.attentions:
            _transformer_shapes(a, d)
    if spec.has_out:
        d["conv_norm_out.weight"] = (boc[0],)
        d["conv_norm_out.bias"] = (boc[0],)
        d["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
        d["conv_out.bias"] = (cfg["out_channels"],)
    return d


def reference_block_order(spec: UNetSpec, fusion_blocks="midup"):
    """BasicTransformerBlock pairing order of ReferenceAttentionControl: torch_dfs(mid)+torch_dfs(up)
    (or the whole unet for 'full'), stable-sorted by descending width
    (magicanimate/models/mutual_self_attention.py:532-543,585-586)."""
    assert fusion_blocks in ("midup", "full")
    items = []
    if fusion_blocks == "full":
        for b in spec.down:
            items += [(a.prefix, a.channels) for a in b.attentions if a is not None]
    # torch_dfs(unet) visits down_blocks, up_blocks, mid_block (module registration order of the
    # reference ctor: mid_block is assigned after both ModuleLists); 'midup' lists mid first.
    if fusion_blocks == "midup":
        items += [(a.prefix, a.channels) for a in spec.mid.attentions]
    for b in spec.up:
        items += [(a.prefix, a.channels) for a in b.attentions if a is not None]
    if fusion_blocks == "full":
        items += [(a.prefix, a.channels) for a in spec.mid.attentions]
    return [p for p, _ in sorted(items, key=lambda it: -it[1])]
