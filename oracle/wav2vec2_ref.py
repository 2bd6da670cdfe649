"""Oracle (test infrastructure): functional fp32 CPU restatement of the wav2vec2 encoder behind the reference's audio front-end.

The reference loads it as a THIRD-PARTY network: `Wav2Vec2Model.from_pretrained('facebook/wav2vec2-base-960h')`
(/root/reference/Net.py:607-612) and consumes `model(input_values).last_hidden_state` (Net.py:643-644).  The algorithm is not in
the reference tree; it lives in the `transformers` package (requirements.txt names it unpinned; the build container holds
transformers - the version is recorded in tests/golden/wav2vec2.json by the generator).  Restated here from its published
architecture (Baevski et al. 2020, "wav2vec 2.0", base configuration: feat_extract_norm="group", do_stable_layer_norm=False):

  feature encoder  7 x Conv1d (no bias): k = (10,3,3,3,3,2,2), stride = (5,2,2,2,2,2,2), 512 channels; layer 0 is followed by
                   GroupNorm(512 groups over 512 channels = per-channel statistics over time) before the GELU, the others by GELU
  projection       LayerNorm(512) -> Linear(512 -> 768)
  encoder          x + GELU(grouped Conv1d(768, 768, k=128, pad=64, groups=16, weight-normalised over dim 2)[..., :-1]) -> LayerNorm ->
                   12 post-LN layers: x = LN(x + MHA(x)) (12 heads of 64, q scaled by 64^-0.5, biased projections),
                   x = LN(x + W2 GELU(W1 x))

PINNED by tests/golden/wav2vec2.safetensors: outputs of transformers' own `Wav2Vec2Model` (random-init, name-keyed synthetic
weights) on seeded waveforms - tools/oracle/gen_golden_wav2vec2.py, tests/test_oracle_golden.py::test_wav2vec2_*.
State-dict keys are transformers' (both spellings of the weight-normalised positional conv are accepted).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BASE_CONFIG = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   conv_dim=(512, 512, 512, 512, 512, 512, 512), conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2),
                   conv_bias=False, feat_extract_norm="group", num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                   do_stable_layer_norm=False, layer_norm_eps=1e-5)


def pos_conv_weight(sd, prefix="encoder.pos_conv_embed.conv"):
    """weight_norm(conv, dim=2): w = g * v / ||v|| with the norm over every dim but 2 (g: (1, 1, k))."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    g = sd.get(prefix + ".parametrizations.weight.original0", sd.get(prefix + ".weight_g"))
    v = sd.get(prefix + ".parametrizations.weight.original1", sd.get(prefix + ".weight_v"))
    return v * (g / v.norm(dim=(0, 1), keepdim=True))


def wav2vec2_forward(sd, cfg, input_values):
    """input_values (B, n_samples) -> last_hidden_state (B, T, hidden).  Eval mode: no dropout, no SpecAugment masking."""
    c = dict(BASE_CONFIG, **(cfg or {}))
    if c["feat_extract_norm"] != "group" or c["do_stable_layer_norm"]:
        raise NotImplementedError("only the wav2vec2-base family (group norm, post-LN) is restated")
    eps = c["layer_norm_eps"]
    h = input_values[:, None]                                                     # (B, 1, n)
    for i, (k, s) in enumerate(zip(c["conv_kernel"], c["conv_stride"])):
        p = f"feature_extractor.conv_layers.{i}"
        h = F.conv1d(h, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), stride=s)
        if i == 0:
            C0 = h.shape[1]
            h = F.group_norm(h, C0, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5)
        h = F.gelu(h)
    h = h.transpose(1, 2)                                                         # (B, T, 512)
    h = F.layer_norm(h, h.shape[-1:], sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], eps)
    h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    kpos = c["num_conv_pos_embeddings"]
    pos = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"], padding=kpos // 2,
                   groups=c["num_conv_pos_embedding_groups"])
    if kpos % 2 == 0:
        pos = pos[:, :, :-1]
    h = h + F.gelu(pos).transpose(1, 2)
    h = F.layer_norm(h, h.shape[-1:], sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], eps)
    heads = c["num_attention_heads"]
    B, T, D = h.shape
    d = D // heads
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layers.{i}"
        lin = lambda n, x: F.linear(x, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"])
        sp = lambda t: t.reshape(B, T, heads, d).permute(0, 2, 1, 3)
        q, k_, v = sp(lin("attention.q_proj", h) * d ** -0.5), sp(lin("attention.k_proj", h)), sp(lin("attention.v_proj", h))
        a = torch.matmul(torch.matmul(q, k_.transpose(-1, -2)).softmax(-1), v).permute(0, 2, 1, 3).reshape(B, T, D)
        h = h + lin("attention.out_proj", a)
        h = F.layer_norm(h, (D,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], eps)
        h = h + lin("feed_forward.output_dense", F.gelu(lin("feed_forward.intermediate_dense", h)))
        h = F.layer_norm(h, (D,), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], eps)
    return h


def normalize_waveform(x):
    """Wav2Vec2FeatureExtractor(do_normalize=True), which `self.processor(waveform, ...)` applies (Net.py:639): zero mean, unit
    variance per utterance, (x - mean) / sqrt(var + 1e-7)."""
    x = x.float()
    return (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-7)


def audio_features(sd, cfg, waveform, m=2, n=2):
    """Wav2VecFeatureExtractor.extract_features_from_wav behind the file read (Net.py:636-667): normalise, encoder, windows of
    frames [f - m, f + n] zero-padded at the ends -> (T, (m + n + 1) * D)."""
    hs = wav2vec2_forward(sd, cfg, normalize_waveform(waveform.reshape(1, -1)))[0]
    T, D = hs.shape
    pad = torch.cat([torch.zeros(m, D), hs, torch.zeros(n, D)])
    return torch.stack([pad[f:f + m + n + 1].flatten() for f in range(T)])
