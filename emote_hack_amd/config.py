"""UNet config surface: the reference ctor's kwargs with the reference's defaults
(magicanimate/models/unet_controlnet.py:57-105) and the YAML loading rules SURVEY.md section 8(b)
lists for `configs/unet-config.yaml` (2D->3D block names, unknown keys dropped, numeric
strings cast - PyYAML reads `1e-05` as a str where the reference's OmegaConf yields a float).
"""
from __future__ import annotations

UNET_DEFAULTS = dict(
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

MOTION_DEFAULTS = dict(  # magicanimate/models/motion_module.py:54-65
    num_attention_heads=8, num_transformer_block=2, attention_block_types=("Temporal_Self", "Temporal_Self"),
    cross_frame_attention_mode=None, temporal_position_encoding=False,
    temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, zero_initialize=True,
)

_DOWN_OK = ("DownBlock3D", "CrossAttnDownBlock3D")
_UP_OK = ("UpBlock3D", "CrossAttnUpBlock3D")


class FrozenConfig(dict):
    """`.config.<kwarg>` attribute access (the pipeline reads unet.config.sample_size)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def normalize_unet_config(kwargs: dict, *, strict: bool = True) -> FrozenConfig:
    """Merge kwargs over the reference defaults and validate like the reference ctor does.
    strict=True rejects unknown kwargs (a Python ctor would raise TypeError)."""
    cfg = dict(UNET_DEFAULTS)
    for k, v in kwargs.items():
        if k not in cfg:
            if strict:
                raise TypeError(f"UNet3DConditionModel.__init__() got an unexpected keyword argument '{k}'")
            continue
        cfg[k] = v
    cfg["norm_eps"] = float(cfg["norm_eps"])
    cfg["block_out_channels"] = tuple(int(c) for c in cfg["block_out_channels"])
    n = len(cfg["block_out_channels"])
    hd = cfg["attention_head_dim"]
    cfg["attention_head_dim"] = tuple(int(h) for h in hd) if isinstance(hd, (list, tuple)) else (int(hd),) * n
    cfg["down_block_types"] = tuple(cfg["down_block_types"])
    cfg["up_block_types"] = tuple(cfg["up_block_types"])
    cfg["motion_module_resolutions"] = tuple(cfg["motion_module_resolutions"])
    for t in cfg["down_block_types"]:  # unet_3d_blocks.py:103
        t2 = t[7:] if t.startswith("UNetRes") else t
        if t2 not in _DOWN_OK:
            raise ValueError(f"{t2} does not exist.")
    for t in cfg["up_block_types"]:  # unet_3d_blocks.py:178
        t2 = t[7:] if t.startswith("UNetRes") else t
        if t2 not in _UP_OK:
            raise ValueError(f"{t2} does not exist.")
    if cfg["mid_block_type"] != "UNetMidBlock3DCrossAttn":  # unet_controlnet.py:203
        raise ValueError(f"unknown mid_block_type : {cfg['mid_block_type']}")
    if cfg["dual_cross_attention"]:
        raise NotImplementedError("dual_cross_attention")  # unet_3d_blocks.py:232
    if cfg["resnet_time_scale_shift"] != "default":
        raise NotImplementedError("resnet_time_scale_shift != 'default' is outside the hot path")
    if cfg["class_embed_type"] not in (None, "timestep", "identity"):      # unet_controlnet.py:120-127: anything else leaves class_embedding None
        cfg["class_embed_type"] = None
    if cfg["unet_use_cross_frame_attention"]:
        raise NotImplementedError("SparseCausalAttention2D is undefined in the reference (attention.py:190)")
    if cfg["unet_use_temporal_attention"]:
        raise NotImplementedError("attn_temp path (off in configs/inference.yaml:3)")
    if cfg["use_motion_module"] and cfg["motion_module_type"] != "Vanilla":
        raise ValueError("motion_module_type must be 'Vanilla'")  # motion_module.py:47-50
    mm = dict(MOTION_DEFAULTS)
    mm.update(cfg["motion_module_kwargs"] or {})
    mm["attention_block_types"] = tuple(mm["attention_block_types"])
    for b in mm["attention_block_types"]:
        if b != "Temporal_Self":
            raise NotImplementedError(f"attention block type {b}")
    cfg["motion_module_kwargs"] = mm
    return FrozenConfig(cfg)


def unet_config_from_yaml(path: str, section: str = "denoising_unet_config", variant: str = "default", **extra):
    """Read `configs/unet-config.yaml` the way the reference's from_config would:
    (i) *2D block names -> *3D, (ii) unknown keys dropped, (iii) norm_num_groups honoured,
    (iv) numeric strings cast."""
    import yaml

    with open(path) as f:
        raw = yaml.safe_load(f)[section][variant]
    raw = dict(raw)
    raw["down_block_types"] = [t.replace("2D", "3D") for t in raw["down_block_types"]]
    raw["up_block_types"] = [t.replace("2D", "3D") for t in raw["up_block_types"]]
    kept = {k: v for k, v in raw.items() if k in UNET_DEFAULTS}
    for k in ("norm_eps", "mid_block_scale_factor"):
        if k in kept:
            kept[k] = float(kept[k])
    kept.setdefault("unet_use_cross_frame_attention", False)
    kept.setdefault("unet_use_temporal_attention", False)
    kept.update(extra)
    return normalize_unet_config(kept)
