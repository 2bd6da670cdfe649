"""The audio front-end of the reference on MI355X: the wav2vec2 encoder and the feature extractor around it.

  Wav2Vec2Model           the network `Wav2VecFeatureExtractor.__init__` loads (Net.py:607-612: transformers'
                          `Wav2Vec2Model.from_pretrained('facebook/wav2vec2-base-960h')`) - third-party, weights from the network,
                          so weights are CALLER-LOADED here (`load_state_dict` takes transformers' key names); the forward runs on
                          the kernels the UNet already uses:
                            feature encoder   7 strided Conv1d layers = GEMMs over an overlapping ROW VIEW of the (time, channel)
                                              activation (window t of kernel k / stride s is k*C contiguous elements at row t*s:
                                              lda = s*C, K = k*C - no im2col copy), emo_channelnorm (+GELU) behind layer 0, emo_act GELU
                            projection        emo_layernorm + emo_gemm
                            positional conv   16 per-group GEMMs over the same kind of row view (k = 128), GELU, residual
                            12 layers         emo_gemm (q | k fused, V^T by the transposed store) + emo_attention (12 heads of 64) +
                                              emo_layernorm + GELU feed-forward
  Wav2VecFeatureExtractor extract_features_from_wav behind the file read (Net.py:636-667): utterance normalisation (what
                          `self.processor(...)` does), encoder, emo_audio_windows -> (T, (m + n + 1) * 768)

Oracle: oracle/wav2vec2_ref.py, pinned by outputs of transformers' own model (tests/golden/wav2vec2.safetensors).
Only the wav2vec2-base family is built (feat_extract_norm "group", post-LN encoder) - the one the reference names.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops
from ._lib import EmoHipError
from .synth import synth_state_dict, synth_tensor

BASE_CONFIG = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   conv_dim=(512, 512, 512, 512, 512, 512, 512), conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2),
                   conv_bias=False, feat_extract_norm="group", num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                   do_stable_layer_norm=False, layer_norm_eps=1e-5)


def _r8(x):
    return (x + 7) // 8 * 8


def wav2vec2_param_shapes(cfg=None):
    """transformers' Wav2Vec2Model state-dict keys / shapes for a config of the base family (parametrised weight-norm spelling)."""
    c = dict(BASE_CONFIG, **(cfg or {}))
    d, cin = {}, 1
    for i, (co, k) in enumerate(zip(c["conv_dim"], c["conv_kernel"])):
        d[f"feature_extractor.conv_layers.{i}.conv.weight"] = (co, cin, k)
        if i == 0:
            d["feature_extractor.conv_layers.0.layer_norm.weight"] = (co,)
            d["feature_extractor.conv_layers.0.layer_norm.bias"] = (co,)
        cin = co
    H, I = c["hidden_size"], c["intermediate_size"]
    d["feature_projection.layer_norm.weight"] = (cin,)
    d["feature_projection.layer_norm.bias"] = (cin,)
    d["feature_projection.projection.weight"] = (H, cin)
    d["feature_projection.projection.bias"] = (H,)
    kp, g = c["num_conv_pos_embeddings"], c["num_conv_pos_embedding_groups"]
    d["encoder.pos_conv_embed.conv.bias"] = (H,)
    d["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, kp)
    d["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (H, H // g, kp)
    d["encoder.layer_norm.weight"] = (H,)
    d["encoder.layer_norm.bias"] = (H,)
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layers.{i}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            d[f"{p}.attention.{n}.weight"] = (H, H)
            d[f"{p}.attention.{n}.bias"] = (H,)
        d[f"{p}.layer_norm.weight"] = (H,)
        d[f"{p}.layer_norm.bias"] = (H,)
        d[f"{p}.feed_forward.intermediate_dense.weight"] = (I, H)
        d[f"{p}.feed_forward.intermediate_dense.bias"] = (I,)
        d[f"{p}.feed_forward.output_dense.weight"] = (H, I)
        d[f"{p}.feed_forward.output_dense.bias"] = (H,)
        d[f"{p}.final_layer_norm.weight"] = (H,)
        d[f"{p}.final_layer_norm.bias"] = (H,)
    return d


def wav2vec2_synth_state_dict(cfg=None, prefix="wav2vec2.", device="cpu"):
    """Name-keyed synthetic weights (emote_hack_amd.synth) for the encoder; the weight-norm magnitude g is drawn around 3 so that the
    positional convolution is numerically visible (a fan-in scaled g would leave only its bias)."""
    sd = synth_state_dict(wav2vec2_param_shapes(cfg), prefix=prefix, device=device)
    k = "encoder.pos_conv_embed.conv.parametrizations.weight.original0"
    sd[k] = 3.0 * synth_tensor(prefix + k + ".g", sd[k].shape[-1:], device=device).reshape(sd[k].shape)     # 3 * (1 + 0.1 N(0,1))
    return sd


class Wav2Vec2Model:
    """forward(input_values (1, n_samples) f32) -> namespace(last_hidden_state=(1, T, hidden) f32), like the transformers class the
    reference calls (Net.py:643-644)."""

    def __init__(self, config=None, **kwargs):
        cfg = dict(BASE_CONFIG)
        if config is not None:
            cfg.update({k: getattr(config, k) for k in BASE_CONFIG if hasattr(config, k)} if not isinstance(config, dict) else config)
        cfg.update(kwargs)
        if cfg["feat_extract_norm"] != "group" or cfg["do_stable_layer_norm"] or cfg["conv_bias"]:
            raise NotImplementedError("Wav2Vec2Model: only the wav2vec2-base family (feat_extract_norm='group', conv_bias=False, "
                                      "do_stable_layer_norm=False) - the checkpoint the reference names (Net.py:608)")
        H, nh = cfg["hidden_size"], cfg["num_attention_heads"]
        if H % nh or (H // nh) % 8 or (H // cfg["num_conv_pos_embedding_groups"]) % 8 or any(c % 8 for c in cfg["conv_dim"]):
            raise ValueError("Wav2Vec2Model: head dim, positional-conv group width and conv channels must be multiples of 8")
        self.config = SimpleNamespace(**cfg)
        self._cfg = cfg
        self._shapes = wav2vec2_param_shapes(cfg)
        self._sd, self._w = None, None
        self.dtype, self.device = torch.float32, torch.device("cpu")

    # ---- torch-module-like surface
    def eval(self):
        return self

    def state_dict(self):
        if self._sd is None:
            raise EmoHipError("no weights loaded")
        return dict(self._sd)

    def load_state_dict(self, sd, strict=True):
        sd = dict(sd)
        pc = "encoder.pos_conv_embed.conv."
        for old, new in ((pc + "weight_g", pc + "parametrizations.weight.original0"), (pc + "weight_v", pc + "parametrizations.weight.original1")):
            if old in sd and new not in sd:       # checkpoints written before torch's parametrised weight_norm
                sd[new] = sd.pop(old)
        missing = [k for k in self._shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._shapes and k != "masked_spec_embed"]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: {len(missing)} missing {missing[:4]}, {len(unexpected)} unexpected {unexpected[:4]}")
        for k, shp in self._shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shp)}")
        self._sd = {k: sd[k].detach().float() for k in self._shapes if k in sd}
        self._pack()
        return missing, unexpected

    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            else:
                device = torch.device(a)
        if dtype is not None:
            self.dtype = dtype
        if device is not None:
            self.device = torch.device(device)
        self._pack()
        return self

    def _pack(self):
        if self._sd is None or self.device.type != "cuda" or any(k not in self._sd for k in self._shapes):
            return
        c, dev, dtp, sd = self._cfg, self.device, self.dtype, self._sd
        w = {}
        f32 = lambda k: sd[k].to(dev).float().contiguous()
        lin = lambda k: sd[k].to(dev, dtp).contiguous()
        for i, k in enumerate(c["conv_kernel"]):
            t = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]              # (Cout, Cin, k) -> [Cout][k][Cin]
            t = t.permute(0, 2, 1).reshape(t.shape[0], -1)
            if i == 0:       # Cin = 1: the k taps padded to a 16-byte multiple (the window matrix is padded alike)
                t = torch.nn.functional.pad(t, (0, _r8(k) - k))
            w[f"conv{i}"] = t.to(dev, dtp).contiguous()
        w["gn.g"], w["gn.b"] = f32("feature_extractor.conv_layers.0.layer_norm.weight"), f32("feature_extractor.conv_layers.0.layer_norm.bias")
        w["fp.ln.g"], w["fp.ln.b"] = f32("feature_projection.layer_norm.weight"), f32("feature_projection.layer_norm.bias")
        w["fp.w"], w["fp.b"] = lin("feature_projection.projection.weight"), f32("feature_projection.projection.bias")
        # weight_norm(dim=2): w = g * v / ||v||, the norm over every dim but 2 - formed once at load in f32
        g = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
        v = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
        wp = v * (g / v.norm(dim=(0, 1), keepdim=True))                           # (H, H/groups, k)
        G = c["num_conv_pos_embedding_groups"]
        cg = wp.shape[0] // G
        w["pos.w"] = [wp[gi * cg:(gi + 1) * cg].permute(0, 2, 1).reshape(cg, -1).to(dev, dtp).contiguous() for gi in range(G)]
        w["pos.b"] = [f32("encoder.pos_conv_embed.conv.bias")[gi * cg:(gi + 1) * cg].contiguous() for gi in range(G)]
        w["enc.ln.g"], w["enc.ln.b"] = f32("encoder.layer_norm.weight"), f32("encoder.layer_norm.bias")
        for i in range(c["num_hidden_layers"]):
            p = f"encoder.layers.{i}"
            w[f"{i}.qk.w"] = torch.cat([sd[f"{p}.attention.q_proj.weight"], sd[f"{p}.attention.k_proj.weight"]]).to(dev, dtp).contiguous()
            w[f"{i}.qk.b"] = torch.cat([sd[f"{p}.attention.q_proj.bias"], sd[f"{p}.attention.k_proj.bias"]]).to(dev).float().contiguous()
            w[f"{i}.v.w"], w[f"{i}.v.b"] = lin(f"{p}.attention.v_proj.weight"), f32(f"{p}.attention.v_proj.bias")
            w[f"{i}.o.w"], w[f"{i}.o.b"] = lin(f"{p}.attention.out_proj.weight"), f32(f"{p}.attention.out_proj.bias")
            w[f"{i}.ln1.g"], w[f"{i}.ln1.b"] = f32(f"{p}.layer_norm.weight"), f32(f"{p}.layer_norm.bias")
            w[f"{i}.f1.w"], w[f"{i}.f1.b"] = lin(f"{p}.feed_forward.intermediate_dense.weight"), f32(f"{p}.feed_forward.intermediate_dense.bias")
            w[f"{i}.f2.w"], w[f"{i}.f2.b"] = lin(f"{p}.feed_forward.output_dense.weight"), f32(f"{p}.feed_forward.output_dense.bias")
            w[f"{i}.ln2.g"], w[f"{i}.ln2.b"] = f32(f"{p}.final_layer_norm.weight"), f32(f"{p}.final_layer_norm.bias")
        self._w = w

    # ---- forward
    @staticmethod
    def _windows(x, k, s):
        """The conv's window matrix as an overlapping row view: x (T, C) contiguous -> (T_out, k*C) with lda = s*C (no copy)."""
        T, C = x.shape
        return torch.as_strided(x, ((T - k) // s + 1, k * C), (s * C, 1))

    @torch.no_grad()
    def forward(self, input_values, attention_mask=None, **_ignored):
        if self._w is None:
            raise EmoHipError("Wav2Vec2Model: load_state_dict + .to('cuda') first (weights are caller-loaded; there is no CPU execution path)")
        if attention_mask is not None:
            raise NotImplementedError("attention_mask: wav2vec2-base-960h is used unmasked (its processor returns no mask; Net.py:639-644)")
        if input_values.dim() != 2 or input_values.shape[0] != 1:
            raise ValueError("input_values must be (1, n_samples): one utterance per call (Net.py:639)")
        c, w, dtp, dev = self._cfg, self._w, self.dtype, self.device
        eps = c["layer_norm_eps"]
        wave = input_values[0].to(dev).float()
        k0, s0 = c["conv_kernel"][0], c["conv_stride"][0]
        if wave.numel() < k0:
            raise ValueError("input_values shorter than the first convolution's kernel")
        # layer 0 (Cin = 1): the (T0, k0) window matrix, padded to 16-byte rows (index / pad only)
        a0 = torch.nn.functional.pad(wave.unfold(0, k0, s0), (0, _r8(k0) - k0)).contiguous()
        h = ops.gemm(ops.convert(a0, dtp), w["conv0"])
        h = ops.channel_norm(h, w["gn.g"], w["gn.b"], 1e-5, gelu=True)            # GroupNorm(C, C) over time, then GELU
        for i in range(1, len(c["conv_kernel"])):
            k, s = c["conv_kernel"][i], c["conv_stride"][i]
            if h.shape[0] < k:
                raise ValueError("input_values too short for the feature encoder")
            h = ops.act(ops.gemm(self._windows(h, k, s), w[f"conv{i}"]), "gelu")
        T = h.shape[0]
        h = ops.gemm(ops.layer_norm(h, w["fp.ln.g"], w["fp.ln.b"], eps), w["fp.w"], w["fp.b"])          # (T, H)
        H = h.shape[1]
        # positional convolution: k = 128, pad 64, groups of H/G channels; the last output step is dropped (even kernel)
        kp, G = c["num_conv_pos_embeddings"], c["num_conv_pos_embedding_groups"]
        cg = H // G
        xp = torch.zeros(G, T + 2 * (kp // 2), cg, device=dev, dtype=dtp)
        xp[:, kp // 2:kp // 2 + T] = h.view(T, G, cg).permute(1, 0, 2)                                  # group-major, zero-padded in time (copy)
        pos = torch.empty(T, H, device=dev, dtype=dtp)
        n_out = T          # (an even kernel yields T + 1 steps, the last one dropped; an odd one yields T)
        for gi in range(G):
            ops.gemm(torch.as_strided(xp[gi], (n_out, kp * cg), (cg, 1)), w["pos.w"][gi], w["pos.b"][gi], out=pos[:, gi * cg:(gi + 1) * cg])
        h = ops.layer_norm(ops.add(h, ops.act(pos, "gelu")), w["enc.ln.g"], w["enc.ln.b"], eps)
        nh = c["num_attention_heads"]
        d = H // nh
        for i in range(c["num_hidden_layers"]):
            qk = ops.gemm(h, w[f"{i}.qk.w"], w[f"{i}.qk.b"])
            vt = ops.gemm(h, w[f"{i}.v.w"], w[f"{i}.v.b"], transpose_rows=T, transpose_ld=_r8(T))
            att = ops.attention(qk[:, :H], qk[:, H:], vt, T, B=1, Lq=T, heads=nh, d=d, scale=d ** -0.5)
            h = ops.layer_norm(ops.gemm(att, w[f"{i}.o.w"], w[f"{i}.o.b"], residual=h), w[f"{i}.ln1.g"], w[f"{i}.ln1.b"], eps)
            f = ops.act(ops.gemm(h, w[f"{i}.f1.w"], w[f"{i}.f1.b"]), "gelu")
            h = ops.layer_norm(ops.gemm(f, w[f"{i}.f2.w"], w[f"{i}.f2.b"], residual=h), w[f"{i}.ln2.g"], w[f"{i}.ln2.b"], eps)
        return SimpleNamespace(last_hidden_state=ops.convert(h, torch.float32).unsqueeze(0))

    __call__ = forward


def normalize_waveform(x: torch.Tensor) -> torch.Tensor:
    """What `self.processor(waveform, sampling_rate=..., return_tensors='pt').input_values` does for wav2vec2-base-960h (Net.py:639;
    Wav2Vec2FeatureExtractor do_normalize=True): zero mean, unit variance per utterance, (x - mean) / sqrt(var + 1e-7).  Host-side
    input preparation, like the processor's numpy code."""
    x = x.float().reshape(1, -1)
    return (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-7)


class Wav2VecFeatureExtractor:
    """Net.py:607-667 without the file IO: `model` is a Wav2Vec2Model with caller-loaded weights (the reference downloads
    'facebook/wav2vec2-base-960h'); `extract_features(waveform)` takes the mono 16 kHz samples `sf.read` / `librosa.resample` produce
    (Net.py:627-636) and returns what extract_features_from_wav returns: (T, (m + n + 1) * hidden) f32."""
    sampling_rate = 16000

    def __init__(self, model: Wav2Vec2Model, device="cuda"):
        self.model, self.device = model, torch.device(device)

    def extract_features(self, waveform, m: int = 2, n: int = 2) -> torch.Tensor:
        wf = torch.as_tensor(waveform, dtype=torch.float32)
        if wf.dim() > 1:
            wf = wf.mean(dim=1)                                    # multi-channel audio: mean over channels (Net.py:634-636)
        hs = self.model(normalize_waveform(wf).to(self.device)).last_hidden_state
        from .conditioning import audio_windows
        return audio_windows(hs, m, n)

    extract_features_from_wav = extract_features
