"""EMO conditioning modules on MI355X - host mirrors of the reference's small
pure-torch classes, same constructor arguments and state-dict keys, arithmetic in HIP kernels.

  SpeedEncoder              Net.py:198-258                 tanh bucket code -> Linear -> ReLU -> Linear
  SpeedController           train_stage_3_speedlayers.py:20-55 (= Net.py:554-589)   argmin bucket (INT) -> Embedding -> MLP
  FaceRegionController      train_stage_3_speedlayers.py:57-76      4x conv3x3 (+ReLU)
  CrossAttentionLayer / AudioAttentionLayers / ReferenceAttentionLayer   Net.py:263-365   single-head attention, q/k/v WITH bias
  AudioAttention / TemporalAttention      train_stage_2_temporal_audio.py:123-177        8-head cross / temporal attention
  stage3_combine            train_stage_3_speedlayers.py:242-271    unet(latents + face) + speed_embed[..., None, None]

None of these is wired into the UNet by the reference (EMOAnimationPipeline.py:783-784 passes kwargs the UNet
rejects); here the UNet accepts their outputs (`speed_embeddings`, `audio_features`).  Reference quirks kept as
documented behaviour: SpeedEncoder has 9 fixed centres (the reference ctor asserts len == num_speed_buckets, so
only 9 works: Net.py:212,221-224); CrossAttentionLayer needs q-len == kv-len only through its assert (:287),
which we keep.
"""
from __future__ import annotations

import math

import torch

from . import ops
from ._lib import EmoHipError


class _HipModule:
    """state-dict driven module: shapes declared by subclasses; weights packed on .to(device, dtype)."""
    _shapes: dict = {}

    def __init__(self):
        self._sd = None
        self.dtype, self.device = torch.float32, torch.device("cpu")
        self._w = None

    def state_dict_shapes(self):
        return dict(self._shapes)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._shapes if k not in sd]
        if strict and missing:
            raise RuntimeError(f"missing keys {missing}")
        for k, shp in self._shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}")
        self._sd = {k: sd[k].detach().float() for k in self._shapes if k in sd}
        self._pack()
        return missing, [k for k in sd if k not in self._shapes]

    def to(self, device=None, dtype=None):
        if dtype is not None:
            self.dtype = dtype
        if device is not None:
            self.device = torch.device(device)
        self._pack()
        return self

    def _pack(self):
        if self._sd is None or self.device.type != "cuda":
            return
        self._w = {k: (v.to(self.device, self.dtype if (v.dim() >= 2 and not k.endswith("temperature")) else torch.float32)).contiguous()
                   for k, v in self._sd.items()}

    def _need(self):
        if self._w is None:
            raise EmoHipError(f"{type(self).__name__}: load_state_dict + .to('cuda') first (no CPU execution path)")

    def _lin(self, x, name, relu=False):
        y = ops.gemm(x, self._w[name + ".weight"], self._w.get(name + ".bias"))
        return ops.act(y, "relu") if relu else y

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class SpeedEncoder(_HipModule):
    CENTERS = [-1.0, -0.5, -0.2, -0.1, 0.0, 0.1, 0.2, 0.5, 1.0]   # Net.py:221-224

    def __init__(self, num_speed_buckets: int, speed_embedding_dim: int):
        super().__init__()
        assert isinstance(num_speed_buckets, int) and num_speed_buckets > 0
        assert isinstance(speed_embedding_dim, int) and speed_embedding_dim > 0
        assert len(self.CENTERS) == num_speed_buckets, "bucket_centers length must match num_speed_buckets"   # Net.py:212
        self.num_speed_buckets, self.speed_embedding_dim = num_speed_buckets, speed_embedding_dim
        D = speed_embedding_dim
        self._shapes = {"mlp.0.weight": (D, num_speed_buckets), "mlp.0.bias": (D,), "mlp.2.weight": (D, D), "mlp.2.bias": (D,)}

    def _pack(self):
        super()._pack()
        if self._w is not None:
            nb = self.num_speed_buckets
            w0 = torch.zeros(self.speed_embedding_dim, 16, device=self.device, dtype=self.dtype)   # K padded 9 -> 16
            w0[:, :nb] = self._w["mlp.0.weight"]
            self._w["mlp.0.weight"] = w0
            self._centers = torch.tensor(self.CENTERS + [0.0] * (16 - nb), device=self.device)
            self._radii = torch.tensor([0.1] * nb + [1e30] * (16 - nb), device=self.device)  # pad: tanh(~0)=0 x zero weight

    def encode_speed(self, v):
        assert v.ndim == 1, "head_rotation_speed must be a 1D tensor"
        self._need()
        return ops.speed_encode(v.to(self.device).float().contiguous(), self._centers, self._radii, self.dtype)[:, :self.num_speed_buckets]

    def forward(self, head_rotation_speeds):
        assert head_rotation_speeds.ndim == 1 and head_rotation_speeds.dtype == torch.float32
        self._need()
        code = ops.speed_encode(head_rotation_speeds.to(self.device).contiguous(), self._centers, self._radii, self.dtype)
        return self._lin(self._lin(code, "mlp.0", relu=True), "mlp.2")


class SpeedController(_HipModule):
    def __init__(self, num_buckets: int = 9, embed_dim: int = 1024):
        super().__init__()
        self.num_buckets, self.embed_dim = num_buckets, embed_dim
        D = embed_dim
        self._shapes = {"speed_embedding.weight": (num_buckets, D), "speed_mlp.0.weight": (D, D), "speed_mlp.0.bias": (D,),
                        "speed_mlp.2.weight": (D, D), "speed_mlp.2.bias": (D,)}

    def map_speed_to_bucket(self, speed):
        centers = torch.linspace(-1.0, 1.0, self.num_buckets).to(self.device)   # constant table, as the reference's buffer
        return ops.speed_bucket(speed.to(self.device).float().contiguous(), centers)

    def forward(self, speeds):
        self._need()
        idx = self.map_speed_to_bucket(speeds)
        e = ops.gather_rows(self._w["speed_embedding.weight"], idx)
        return self._lin(self._lin(e, "speed_mlp.0", relu=True), "speed_mlp.2")


class FaceRegionController(_HipModule):
    def __init__(self, in_channels: int = 1, out_channels: int = 1024):
        super().__init__()
        self.chans = [in_channels, 64, 128, 256, out_channels]
        self._shapes = {}
        for i, k in enumerate((0, 2, 4, 6)):
            self._shapes[f"encoder.{k}.weight"] = (self.chans[i + 1], self.chans[i], 3, 3)
            self._shapes[f"encoder.{k}.bias"] = (self.chans[i + 1],)

    def _pack(self):
        if self._sd is None or self.device.type != "cuda":
            return
        self._w = {}
        for i, k in enumerate((0, 2, 4, 6)):
            t = self._sd[f"encoder.{k}.weight"].to(self.device)
            co, ci = t.shape[:2]
            cip = (ci + 7) // 8 * 8
            o = torch.zeros(co, 3, 3, cip, device=self.device)
            o[..., :ci] = t.permute(0, 2, 3, 1)
            self._w[f"encoder.{k}.weight"] = o.reshape(co, 9 * cip).to(self.dtype).contiguous()
            self._w[f"encoder.{k}.bias"] = self._sd[f"encoder.{k}.bias"].to(self.device).float().contiguous()

    def forward(self, mask):
        """mask (B, Cin, H, W) -> (B, D, H, W) f32."""
        self._need()
        B, Cin, H, W = mask.shape
        x = ops.ncfhw_to_rows(mask.to(self.device).unsqueeze(2), self.dtype, cpad=(Cin + 7) // 8 * 8)
        for i, k in enumerate((0, 2, 4, 6)):
            x, _, _ = ops.conv3x3(x, self._w[f"encoder.{k}.weight"], self._w[f"encoder.{k}.bias"], B, H, W)
            if k != 6:
                x = ops.act(x, "relu")
        return ops.rows_to_ncfhw(x, B, self.chans[-1], 1, H, W).squeeze(2)


class CrossAttentionLayer(_HipModule):
    """Net.py:263-303: single head, q/k/v Linear with bias, scores / sqrt(feature_dim)."""

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim
        D = feature_dim
        self._shapes = {f"{n}.{w}": ((D, D) if w == "weight" else (D,)) for n in ("query", "key", "value") for w in ("weight", "bias")}

    def forward(self, latent_code, audio_features, _prefix=""):
        assert latent_code.dim() == 3 and audio_features.dim() == 3
        assert latent_code.size(1) == audio_features.size(1), "Feature dimensions of latent_code and audio_features must match"
        self._need()
        B, Lq, D = latent_code.shape
        Lk = audio_features.shape[1]
        x = ops.convert(latent_code.to(self.device).float().reshape(-1, D), self.dtype)
        a = ops.convert(audio_features.to(self.device).float().reshape(-1, D), self.dtype)
        w = self._w
        q = ops.gemm(x, w[_prefix + "query.weight"], w[_prefix + "query.bias"])
        k = ops.gemm(a, w[_prefix + "key.weight"], w[_prefix + "key.bias"])
        vt = ops.gemm(a, w[_prefix + "value.weight"], w[_prefix + "value.bias"], transpose_rows=Lk, transpose_ld=(Lk + 7) // 8 * 8)
        o = ops.attention(q, k, vt, Lk, B=B, Lq=Lq, heads=1, d=D, scale=1.0 / math.sqrt(D))
        return ops.convert(o, torch.float32).reshape(B, Lq, D)


class AudioAttentionLayers(_HipModule):
    """Net.py:305-325: latent = layer(latent, audio) + latent per layer."""

    def __init__(self, feature_dim, num_layers):
        super().__init__()
        assert feature_dim > 0 and num_layers > 0
        self.feature_dim, self.num_layers = feature_dim, num_layers
        D = feature_dim
        self._shapes = {f"layers.{i}.{n}.{w}": ((D, D) if w == "weight" else (D,)) for i in range(num_layers)
                        for n in ("query", "key", "value") for w in ("weight", "bias")}
        self._layer = CrossAttentionLayer(feature_dim)

    def forward(self, latent_code, audio_features):
        self._need()
        self._layer._w, self._layer.dtype, self._layer.device = self._w, self.dtype, self.device
        x = latent_code.to(self.device).float()
        for i in range(self.num_layers):
            y = self._layer.forward(x, audio_features, _prefix=f"layers.{i}.")
            x = ops.add(y.reshape(-1, self.feature_dim), x.reshape(-1, self.feature_dim).contiguous()).reshape(x.shape)
        return x


class ReferenceAttentionLayer(CrossAttentionLayer):
    """Net.py:333-365: same single-head attention, residual inside; reference (B, 1, D) key/value."""

    def forward(self, latent_code, reference_features):
        self._need()
        B, Lq, D = latent_code.shape
        Lk = reference_features.shape[1]
        x = ops.convert(latent_code.to(self.device).float().reshape(-1, D), self.dtype)
        a = ops.convert(reference_features.to(self.device).float().reshape(-1, D), self.dtype)
        w = self._w
        q = ops.gemm(x, w["query.weight"], w["query.bias"])
        k = ops.gemm(a, w["key.weight"], w["key.bias"])
        vt = ops.gemm(a, w["value.weight"], w["value.bias"], transpose_rows=Lk, transpose_ld=(Lk + 7) // 8 * 8)
        o = ops.attention(q, k, vt, Lk, B=B, Lq=Lq, heads=1, d=D, scale=1.0 / math.sqrt(D))
        return ops.convert(ops.add(o, x), torch.float32).reshape(B, Lq, D)


class AudioAttention(_HipModule):
    """train_stage_2_temporal_audio.py:146-177: q = frame_proj(x); k = v = audio_proj(a) (shared); 8 heads."""

    def __init__(self, frame_dim: int, audio_dim: int, num_heads: int = 8):
        super().__init__()
        self.frame_dim, self.audio_dim, self.num_heads = frame_dim, audio_dim, num_heads
        self.scale = (frame_dim // num_heads) ** -0.5
        C = frame_dim
        self._shapes = {"frame_proj.weight": (C, C), "frame_proj.bias": (C,), "audio_proj.weight": (C, audio_dim),
                        "audio_proj.bias": (C,), "out_proj.weight": (C, C), "out_proj.bias": (C,)}

    def forward(self, frame_features, audio_features):
        self._need()
        B, T, C = frame_features.shape
        La = audio_features.shape[1]
        x = ops.convert(frame_features.to(self.device).float().reshape(-1, C), self.dtype)
        a = ops.convert(audio_features.to(self.device).float().reshape(-1, self.audio_dim), self.dtype)
        w = self._w
        q = ops.gemm(x, w["frame_proj.weight"], w["frame_proj.bias"])
        k = ops.gemm(a, w["audio_proj.weight"], w["audio_proj.bias"])
        vt = ops.gemm(a, w["audio_proj.weight"], w["audio_proj.bias"], transpose_rows=La, transpose_ld=(La + 7) // 8 * 8)
        o = ops.attention(q, k, vt, La, B=B, Lq=T, heads=self.num_heads, d=C // self.num_heads, scale=self.scale)
        return ops.convert(ops.gemm(o, w["out_proj.weight"], w["out_proj.bias"]), torch.float32).reshape(B, T, C)


class TemporalAttention(_HipModule):
    """train_stage_2_temporal_audio.py:123-144: fused qkv (no bias); scores * per-head `temperature` (init 1.0, NOT
    d^-0.5).  The temperature is folded into the q rows of the packed qkv weight (exact: a per-head scalar)."""

    def __init__(self, dim: int, num_heads: int = 8):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self._shapes = {"temperature": (num_heads, 1, 1), "qkv.weight": (3 * dim, dim), "proj.weight": (dim, dim), "proj.bias": (dim,)}

    def _pack(self):
        if self._sd is None or self.device.type != "cuda":
            return
        C, H = self.dim, self.num_heads
        wq = self._sd["qkv.weight"].clone()
        wq[:C] = wq[:C] * self._sd["temperature"].reshape(H, 1, 1).expand(H, C // H, 1).reshape(C, 1)
        self._w = {"q": wq[:C].to(self.device, self.dtype).contiguous(), "k": wq[C:2 * C].to(self.device, self.dtype).contiguous(),
                   "v": wq[2 * C:].to(self.device, self.dtype).contiguous(),
                   "proj.weight": self._sd["proj.weight"].to(self.device, self.dtype).contiguous(),
                   "proj.bias": self._sd["proj.bias"].to(self.device).float().contiguous()}

    def forward(self, x):
        self._need()
        B, T, C = x.shape
        xr = ops.convert(x.to(self.device).float().reshape(-1, C), self.dtype)
        q, k = ops.gemm(xr, self._w["q"]), ops.gemm(xr, self._w["k"])
        vt = ops.gemm(xr, self._w["v"], transpose_rows=T, transpose_ld=(T + 7) // 8 * 8)
        o = ops.attention(q, k, vt, T, B=B, Lq=T, heads=self.num_heads, d=C // self.num_heads, scale=1.0)
        return ops.convert(ops.gemm(o, self._w["proj.weight"], self._w["proj.bias"]), torch.float32).reshape(B, T, C)


def stage3_combine(unet_fn, noisy_latents, face_features, speed_embed):
    """EMOStage3.forward combine rule (train_stage_3_speedlayers.py:242-271) on 4-D latents (B, C, H, W):
    unet(latents + face_features) + speed_embed[..., None, None].  Elementwise parts run in HIP."""
    B, Cc, H, W = noisy_latents.shape
    dev = face_features.device if face_features.is_cuda else torch.device("cuda")
    a = ops.ncfhw_to_rows(noisy_latents.to(dev).unsqueeze(2), torch.float32)
    f = ops.ncfhw_to_rows(face_features.to(dev).unsqueeze(2), torch.float32)
    aug = ops.rows_to_ncfhw(ops.add(a, f), B, Cc, 1, H, W).squeeze(2)
    out = unet_fn(aug)
    o = ops.ncfhw_to_rows(out.to(dev).unsqueeze(2), torch.float32)
    o = ops.add_rowbias(o, speed_embed.to(dev).float().contiguous(), H * W)
    return ops.rows_to_ncfhw(o, B, out.shape[1], 1, H, W).squeeze(2)


# ----------------------------------------------------------------------------- VideoNet attention modules
class SpatialAttentionModule(_HipModule):
    """models/videonet.py:15-77 - the alternative reference attention: the reference feature map is concatenated to x ALONG THE
    WIDTH, GroupNorm(32, eps 1e-6) + 1x1 conv over the concatenation, q from the RAW x tokens, k / v from the concatenated
    tokens, `num_heads` heads (xformers.memory_efficient_attention = softmax(q k^T d^-0.5) v), LN(attn + x), Linear, LN(. + .),
    1x1 conv, + x.  The residual `attn_out + reshaped_x` needs embed_dim == num_inp_channels (the reference's default 40 only
    runs for 40-channel inputs).  x, reference_tensor: (b*t, C, h, w) -> (b*t, C, h, w) f32."""

    def __init__(self, num_inp_channels: int, embed_dim: int = 40, num_heads: int = 8):
        super().__init__()
        if embed_dim != num_inp_channels:
            raise ValueError("SpatialAttentionModule: attn_out + reshaped_x (videonet.py:66) needs embed_dim == num_inp_channels")
        C, E = num_inp_channels, embed_dim
        self.C, self.E, self.num_heads = C, E, num_heads
        self._shapes = {"norm_in.weight": (C,), "norm_in.bias": (C,), "proj_in.weight": (C, C, 1, 1), "proj_in.bias": (C,),
                        "to_q.weight": (E, C), "to_q.bias": (E,), "to_k.weight": (E, C), "to_k.bias": (E,),
                        "to_v.weight": (E, C), "to_v.bias": (E,), "norm1.weight": (E,), "norm1.bias": (E,),
                        "ffn.weight": (E, E), "ffn.bias": (E,), "norm2.weight": (E,), "norm2.bias": (E,),
                        "proj_out.weight": (C, C, 1, 1), "proj_out.bias": (C,)}

    def _pack(self):
        if self._sd is None or self.device.type != "cuda":
            return
        self._w = {k: (v.reshape(v.shape[0], -1).to(self.device, self.dtype) if v.dim() >= 2 else v.to(self.device).float()).contiguous()
                   for k, v in self._sd.items()}

    def _tail(self, att, xr, n_rows_shape):
        """LN(attn + x) -> Linear -> LN(. + .) -> 1x1 conv -> + x   (videonet.py:66-77)"""
        w = self._w
        n1 = ops.layer_norm(ops.add(att, xr), w["norm1.weight"], w["norm1.bias"])
        f = ops.gemm(n1, w["ffn.weight"], w["ffn.bias"])
        n2 = ops.layer_norm(ops.add(n1, f), w["norm2.weight"], w["norm2.bias"])
        return ops.gemm(n2, w["proj_out.weight"], w["proj_out.bias"], residual=xr)

    def forward(self, x, reference_tensor):
        self._need()
        bt, C, h, wd = x.shape
        w, dtp = self._w, self.dtype
        xr = ops.ncfhw_to_rows(x.to(self.device).float().unsqueeze(2), dtp)                        # (bt*h*w, C)
        rr = ops.ncfhw_to_rows(reference_tensor.to(self.device).float().unsqueeze(2), dtp)
        cat = torch.cat([xr.view(bt, h, wd, C), rr.view(bt, h, wd, C)], dim=2).reshape(bt * h * 2 * wd, C)   # concat along w (:42)
        g = ops.group_norm(cat, w["norm_in.weight"], w["norm_in.bias"], bt, 32, 1e-6, False)
        pj = ops.gemm(g, w["proj_in.weight"], w["proj_in.bias"])
        Lk = h * 2 * wd
        q = ops.gemm(xr, w["to_q.weight"], w["to_q.bias"])
        k = ops.gemm(pj, w["to_k.weight"], w["to_k.bias"])
        vt = ops.gemm(pj, w["to_v.weight"], w["to_v.bias"], transpose_rows=Lk, transpose_ld=(Lk + 7) // 8 * 8)
        d = self.E // self.num_heads
        att = ops.attention(q, k, vt, Lk, B=bt, Lq=h * wd, heads=self.num_heads, d=d, scale=d ** -0.5)
        out = self._tail(att, xr, None)
        return ops.rows_to_ncfhw(out, bt, C, 1, h, wd)[:, :, 0]


class TemporalAttentionModule(SpatialAttentionModule):
    """models/videonet.py:81-128: per pixel, ONE head of embed_dim over the `num_frames` frames; q / k / v from the RAW tokens
    (the module computes norm_in / proj_in and never uses the result, :110-118 - neither do we), then the same
    LN / Linear / LN / 1x1-conv tail, + x.  x: ((b t), C, h, w)."""

    def __init__(self, num_inp_channels: int, num_frames: int, embed_dim: int = 40, num_heads: int = 8):
        super().__init__(num_inp_channels, embed_dim, num_heads)
        self.num_frames = num_frames

    def _pack(self):
        super()._pack()
        if self._w is not None:
            self._w["qkv.weight"] = torch.cat([self._w["to_q.weight"], self._w["to_k.weight"], self._w["to_v.weight"]]).contiguous()
            self._w["qkv.bias"] = torch.cat([self._w["to_q.bias"], self._w["to_k.bias"], self._w["to_v.bias"]]).contiguous()

    def forward(self, x):
        self._need()
        bt, C, h, wd = x.shape
        T = self.num_frames
        xr = ops.ncfhw_to_rows(x.to(self.device).float().reshape(bt // T, T, C, h, wd).permute(0, 2, 1, 3, 4), self.dtype)   # ((b t) h w, C)
        qkv = ops.gemm(xr, self._w["qkv.weight"], self._w["qkv.bias"])
        att = ops.temporal_attention(qkv, bt // T, T, h * wd, 1, self.E, self.E ** -0.5)          # 3-D xformers call: one head
        out = self._tail(att, xr, None)
        return ops.rows_to_ncfhw(out, bt // T, C, T, h, wd).permute(0, 2, 1, 3, 4).reshape(bt, C, h, wd)


# ----------------------------------------------------------------------------- FaceLocator
class FaceLocator(_HipModule):
    """Net.py:819-855: conv3x3(3->16) ReLU pool, conv3x3(16->32) ReLU pool, conv3x3(32->64) ReLU pool, 1x1 conv (64->1),
    bilinear upsampling of the logits to the input size (align_corners=False).  images (B, 3, H, W) f32 -> logits (B, 1, H, W);
    H, W multiples of 8.  The reference asserts float32 4-D input (:832-835); so does this."""

    def __init__(self):
        super().__init__()
        self._shapes = {"conv1.weight": (16, 3, 3, 3), "conv1.bias": (16,), "conv2.weight": (32, 16, 3, 3), "conv2.bias": (32,),
                        "conv3.weight": (64, 32, 3, 3), "conv3.bias": (64,), "final_conv.weight": (1, 64, 1, 1), "final_conv.bias": (1,)}

    def _pack(self):
        if self._sd is None or self.device.type != "cuda":
            return
        w = {}
        for k, t in self._sd.items():
            if k.endswith(".bias"):
                w[k] = t.to(self.device).float().contiguous()
            elif t.shape[-1] == 3:
                co, ci = t.shape[0], t.shape[1]
                cip = (ci + 7) // 8 * 8
                o = torch.zeros(co, 3, 3, cip)
                o[..., :ci] = t.permute(0, 2, 3, 1)
                w[k] = o.reshape(co, 9 * cip).to(self.device, self.dtype).contiguous()
            else:
                w[k] = t.reshape(t.shape[0], -1).to(self.device, self.dtype).contiguous()
        self._w = w

    def forward(self, images):
        self._need()
        assert images.dtype == torch.float32, "Images must be of type torch.float32"
        assert images.ndim == 4, "Images must have 4 dimensions [B, C, H, W]"
        B, _, H, W = images.shape
        if H % 8 or W % 8:
            raise ValueError("FaceLocator: H and W must be multiples of 8 (three 2x2 poolings)")
        w = self._w
        x = ops.ncfhw_to_rows(images.to(self.device).permute(1, 0, 2, 3).unsqueeze(0), self.dtype, cpad=8)
        h_, w_ = H, W
        for name in ("conv1", "conv2", "conv3"):
            x, _, _ = ops.conv3x3(x, w[name + ".weight"], w[name + ".bias"], B, h_, w_)
            x = ops.maxpool2x2(ops.act(x, "relu"), B, h_, w_)
            h_, w_ = h_ // 2, w_ // 2
        logits = ops.gemm(x, w["final_conv.weight"], w["final_conv.bias"])
        return ops.bilinear_to_nchw(logits, B, 1, h_, w_, H, W)


# ----------------------------------------------------------------------------- audio front-end
def audio_windows(hidden_states: torch.Tensor, m: int = 2, n: int = 2) -> torch.Tensor:
    """Wav2VecFeatureExtractor.extract_features_from_wav, the windowing half (Net.py:649-667): for every audio frame f the
    wav2vec2 `last_hidden_state` rows [f - m, f + n], zero-padded where they fall outside the clip, flattened -
    (T, D) or (1, T, D) -> (T, (m + n + 1) * D).  The wav2vec2 encoder itself (third-party `transformers`, weights from the
    network) stays caller-supplied: pass its `last_hidden_state`.  Bit-exact (a gather; emo_audio_windows)."""
    hs = hidden_states[0] if hidden_states.dim() == 3 else hidden_states
    return ops.audio_windows(hs, m, n).reshape(hs.shape[0], -1)


def audio_context_tokens(windows: torch.Tensor, num_video_frames: int, feature_dim: int = 768) -> torch.Tensor:
    """Per-video-frame attn2 context for the UNet (`audio_features=`): windows (T_a, W * D) -> (F, W, D).  The reference never
    aligns the 50 Hz wav2vec frames with the video frames (the audio path is unwired); this picks audio frame
    floor(i * T_a / F) for video frame i (index arithmetic only - a documented design choice, not reference behaviour)."""
    Ta = windows.shape[0]
    idx = torch.div(torch.arange(num_video_frames, device=windows.device) * Ta, num_video_frames, rounding_mode="floor").clamp_(max=Ta - 1)
    return windows.index_select(0, idx).reshape(num_video_frames, -1, feature_dim)
