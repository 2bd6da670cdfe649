"""Net.py's placeholder modules - the never-wired sketches around the Backbone, on the HIP kernels where the reference's
own code runs at all (tests/golden/net_placeholders.{safetensors,json} record what the reference does, including what it raises):

  ReferenceAttention             Net.py:1487-1511   q = Wq LN(x), k = Wk LN(ref), v = Wv ref (no norm), ONE head of `channels`, scale
                                 channels^-0.5, no output projection, no residual.  (B, C, H, W) x 2 -> (B, C, H, W)
  MotionModule + TemporalAttention   :1449-1485     Conv3d(C, C, (k, 1, 1), pad k // 2) over time, then LayerNorm(channels) +
                                 nn.MultiheadAttention(channels, 8 heads) over tokens (T, B, C*H*W), + identity.  LayerNorm(channels) on a
                                 C*H*W-wide axis: the reference runs for 1x1 feature maps and odd kernels only; so does this (ValueError
                                 otherwise, where torch raises RuntimeError)
  TemporalModule                 :520-552           `x.view(b, -1, c, h, w)` feeds Conv3d(channels, ...) a ONE-channel volume: the reference
                                 raises for every constructible instance - NotImplementedError here, nothing to run
  BackboneNetwork                :368-411           reference_net -> ReferenceAttentionLayer stack (+ skip) -> AudioAttentionLayers -> the
                                 VanillaTemporalModule stage, which receives the 3-D latent and trips its ndim == 5 assertion in the reference;
                                 `forward` reproduces that, `forward_before_temporal` returns what the runnable stages compute

The INTENT these sketch (motion frames of the previous clip conditioning the next, temporal modules at every resolution) is realised in the
product path by prepare_denoise(motion_latents=) / denoise_chained and the UNet's motion modules."""
from __future__ import annotations

import torch

from . import ops
from .conditioning import AudioAttentionLayers, ReferenceAttentionLayer, _HipModule


def _r8(x):
    return (x + 7) // 8 * 8


class ReferenceAttention(_HipModule):
    def __init__(self, channels: int):
        super().__init__()
        if channels % 8 or channels > 160:
            raise NotImplementedError("ReferenceAttention: ONE head of `channels` (Net.py:1497) - served for channels % 8 == 0 and <= 160")
        C = self.channels = channels
        self._shapes = {"norm.weight": (C,), "norm.bias": (C,), **{f"{n}.{w}": ((C, C) if w == "weight" else (C,))
                                                                    for n in ("q_proj", "k_proj", "v_proj") for w in ("weight", "bias")}}

    def forward(self, x, ref):
        self._need()
        B, C, H, W = x.shape
        w, dtp = self._w, self.dtype
        xr = ops.ncfhw_to_rows(x.to(self.device).float().unsqueeze(2), dtp)            # x.flatten(2).permute(0, 2, 1)  (:1501)
        rr = ops.ncfhw_to_rows(ref.to(self.device).float().unsqueeze(2), dtp)
        L = ref.shape[2] * ref.shape[3]
        q = ops.gemm(ops.layer_norm(xr, w["norm.weight"], w["norm.bias"]), w["q_proj.weight"], w["q_proj.bias"])
        k = ops.gemm(ops.layer_norm(rr, w["norm.weight"], w["norm.bias"]), w["k_proj.weight"], w["k_proj.bias"])
        vt = ops.gemm(rr, w["v_proj.weight"], w["v_proj.bias"], transpose_rows=L, transpose_ld=_r8(L))      # v from the RAW reference tokens (:1506)
        o = ops.attention(q, k, vt, L, B=B, Lq=H * W, heads=1, d=C, scale=C ** -0.5)
        return ops.rows_to_ncfhw(o, B, C, 1, H, W)[:, :, 0]


class TemporalAttention(_HipModule):
    """Net.py:1470-1485 at H = W = 1 (the only geometry its LayerNorm(channels) accepts): x (B, C, T, 1, 1) -> attention output, no residual."""

    def __init__(self, channels: int):
        super().__init__()
        if channels % 64:
            raise NotImplementedError("TemporalAttention: 8 heads (Net.py:1477) of a multiple of 8 channels")
        C = self.channels = channels
        self._shapes = {"norm.weight": (C,), "norm.bias": (C,), "attention.in_proj_weight": (3 * C, C), "attention.in_proj_bias": (3 * C,),
                        "attention.out_proj.weight": (C, C), "attention.out_proj.bias": (C,)}

    def _rows(self, rows, B, T):
        """rows ((b t), C) -> rows ((b t), C): LN -> MHA over t per batch (nn.MultiheadAttention scales q by d^-0.5) -> out_proj."""
        w, C = self._w, self.channels
        n = ops.layer_norm(rows, w["norm.weight"], w["norm.bias"])
        qkv = ops.gemm(n, w["attention.in_proj_weight"], w["attention.in_proj_bias"])
        vt = ops.gemm(n, w["attention.in_proj_weight"][2 * C:], w["attention.in_proj_bias"][2 * C:].contiguous(), transpose_rows=T, transpose_ld=_r8(T))
        d = C // 8
        o = ops.attention(qkv[:, :C], qkv[:, C:2 * C], vt, T, B=B, Lq=T, heads=8, d=d, scale=d ** -0.5)
        return ops.gemm(o, w["attention.out_proj.weight"], w["attention.out_proj.bias"])

    def forward(self, x):
        self._need()
        B, C, T, H, W = x.shape
        if H * W != 1:
            raise ValueError("TemporalAttention: LayerNorm(channels) over a C*H*W-wide axis (Net.py:1481-1482) only exists for 1x1 feature maps")
        rows = ops.convert(x.to(self.device).float().reshape(B, C, T).permute(0, 2, 1).reshape(B * T, C).contiguous(), self.dtype)
        return ops.convert(self._rows(rows, B, T), torch.float32).reshape(B, T, C).permute(0, 2, 1).reshape(B, C, T, 1, 1)


class MotionModule(_HipModule):
    """Net.py:1449-1468: x + TemporalAttention(Conv3d over time (x)).  x (B, C, T, 1, 1)."""

    def __init__(self, in_channels: int, temporal_length: int):
        super().__init__()
        self.temporal_attention = TemporalAttention(in_channels)
        self.in_channels, self.temporal_length = in_channels, temporal_length
        C, k = in_channels, temporal_length
        self._shapes = {"temporal_conv.weight": (C, C, k, 1, 1), "temporal_conv.bias": (C,),
                        **{"temporal_attention." + n: s for n, s in self.temporal_attention._shapes.items()}}

    def _pack(self):
        if self._sd is None or self.device.type != "cuda":
            return
        C, k = self.in_channels, self.temporal_length
        self._w = {"conv.w": self._sd["temporal_conv.weight"].reshape(C, C, k).permute(0, 2, 1).reshape(C, k * C).to(self.device, self.dtype).contiguous(),
                   "conv.b": self._sd["temporal_conv.bias"].to(self.device).float().contiguous()}
        ta = self.temporal_attention
        ta._sd = {n[len("temporal_attention."):]: v for n, v in self._sd.items() if n.startswith("temporal_attention.")}
        ta.to(self.device, self.dtype)

    def forward(self, x):
        self._need()
        B, C, T, H, W = x.shape
        k = self.temporal_length
        if H * W != 1:
            raise ValueError("MotionModule: its TemporalAttention normalises a C*H*W-wide axis with LayerNorm(channels) (Net.py:1481-1482): 1x1 feature maps only")
        if k % 2 == 0:
            raise ValueError("MotionModule: an even temporal_length yields T + 1 frames, which `x + identity` (Net.py:1467) rejects")
        p = k // 2
        xr = ops.convert(x.to(self.device).float().reshape(B, C, T).permute(0, 2, 1).reshape(B * T, C).contiguous(), self.dtype)
        # the conv over time as ONE GEMM over an overlapping row view of the zero-padded (b, t + 2p, C) rows: window r covers rows r .. r + k;
        # windows that start in a batch's tail padding straddle two batches and are dropped
        xp = torch.zeros(B, T + 2 * p, C, device=self.device, dtype=self.dtype)
        xp[:, p:p + T] = xr.view(B, T, C)
        M = B * (T + 2 * p) - 2 * p
        y = ops.gemm(torch.as_strided(xp, (M, k * C), (C, 1)), self._w["conv.w"], self._w["conv.b"])
        keep = (torch.arange(B, device=self.device)[:, None] * (T + 2 * p) + torch.arange(T, device=self.device)[None, :]).reshape(-1)
        y = y.index_select(0, keep)
        out = ops.add(self.temporal_attention._rows(y, B, T), xr)                                   # + identity (:1467)
        return ops.convert(out, torch.float32).reshape(B, T, C).permute(0, 2, 1).reshape(B, C, T, 1, 1)


class TemporalModule:
    """Net.py:520-552.  `x.view(b, -1, c, h, w)` turns (b, c, h, w) into a ONE-channel volume for Conv3d(channels, channels, ...), and
    nn.MultiheadAttention(channels, num_heads=8) needs channels % 8 == 0: no instance runs (the reference raises RuntimeError,
    tests/golden/net_placeholders.json).  Kept as a name so that code importing it fails with the reason."""

    def __init__(self, channels: int, num_frames: int):
        self.channels, self.num_frames = channels, num_frames

    def forward(self, x, motion_frames=None):
        raise NotImplementedError("TemporalModule.forward does not run in the reference either (Net.py:537-545: a 1-channel view handed to "
                                  "Conv3d(channels, ...)); the product's temporal path is the UNet's motion modules")

    __call__ = forward


class BackboneNetwork:
    """Net.py:368-411.  reference_net: callable ref_image -> reference features (B, 1, D); audio_attention_layers: an AudioAttentionLayers."""

    def __init__(self, feature_dim, num_layers, reference_net, audio_attention_layers, temporal_module_kwargs=None):
        self.feature_dim, self.num_layers = feature_dim, num_layers
        self.reference_net, self.audio_attention_layers = reference_net, audio_attention_layers
        self.reference_attention_layers = [ReferenceAttentionLayer(feature_dim) for _ in range(num_layers)]
        self.temporal_module_kwargs = dict(temporal_module_kwargs or {})

    def load_state_dict(self, sd, strict=True):
        """keys: reference_attention_layers.<i>.{query,key,value}.{weight,bias} (+ temporal_modules.*, which no forward reaches)"""
        for i, layer in enumerate(self.reference_attention_layers):
            p = f"reference_attention_layers.{i}."
            layer.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)}, strict=strict)

    def to(self, device=None, dtype=None):
        for layer in self.reference_attention_layers:
            layer.to(device, dtype)
        return self

    def forward_before_temporal(self, latent_code, audio_features, ref_image):
        reference_features = self.reference_net(ref_image)                                            # :399
        for layer in self.reference_attention_layers:
            latent_code = ops.add(layer(latent_code, reference_features).reshape(-1, self.feature_dim).contiguous(),
                                  latent_code.to(layer.device).float().reshape(-1, self.feature_dim).contiguous()).reshape(latent_code.shape)   # :402-403
        return self.audio_attention_layers(latent_code, audio_features)                               # :406

    def forward(self, latent_code, audio_features, ref_image):
        latent_code = self.forward_before_temporal(latent_code, audio_features, ref_image)
        if self.num_layers > 0:     # :408-409 hands the 3-D latent to VanillaTemporalModule, whose forward starts with this assertion
            assert latent_code.dim() == 5, f"Expected hidden_states to have ndim=5, but got ndim={latent_code.dim()}."
        return latent_code

    __call__ = forward
