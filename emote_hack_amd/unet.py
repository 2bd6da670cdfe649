"""UNet3DConditionModel - the Backbone denoising UNet (and, at F=1 without motion modules, the
ReferenceNet) on MI355X.  Host-side mirror of the reference class
(magicanimate/models/unet_controlnet.py:54-525): same ctor kwargs, `.config`, forward signature,
`UNet3DConditionOutput.sample`, state-dict keys, `from_pretrained_2d`.

Underneath there is no torch arithmetic: forward() is a sequence of hand-written HIP kernel launches
(emote_hack_amd/csrc via the C ABI in include/emo_hip.h) over NHWC "rows x channels" activations.
The (b f) h w token layout of the transformers and the NHWC layout of the convs are the same memory,
so conv -> attention -> temporal attention transitions move no data.

Weights are re-laid once at load (SURVEY.md section 7 "hard parts"):
  conv 3x3  OIHW -> [Cout][ky][kx][Cin]      1x1 / Linear -> [N][K]
  GEGLU proj rows interleaved (32 value, 32 gate) so the gate is applied in the GEMM epilogue
  attn1 to_q|to_k fused; motion-module to_q|to_k|to_v fused; all 22 time_emb_proj fused into one GEMM.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch

from . import ops
from ._lib import EmoHipError
from .config import FrozenConfig, normalize_unet_config
from .spec import UNetSpec, build_spec, param_shapes, reference_block_order


@dataclass
class UNet3DConditionOutput:  # unet_controlnet.py:49-51
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


def _round_up(x, m):
    return (x + m - 1) // m * m


# Every LayerNorm of the transformer / motion blocks feeds a Linear (attention.py:279-316, motion_module.py:216-224,282-303):
# the affine is folded into the Linear's operands at load and the normalisation into the GEMM (emo_gemm_params.ln_colsum /
# ln_stats): a read-only statistics pass replaces LayerNorm's read + write and the GEMM reads the raw tokens.  The one exception is the LN1 of a block
# whose output is WRITTEN to a reference bank (mutual_self_attention.py:230): that tensor has to exist.
FOLD_LAYERNORM = True
# The two halves of a classifier-free-guidance batch share everything up to the first cross-attention (see _begin): computed once.
SHARE_CFG_PREFIX = True
# ff.net.2 and proj_out as ONE GEMM over K = 4C + C: (g W2^T + b2 + h) Wo^T + bo + x = [g | h] [Wo W2 | Wo]^T + (Wo b2 + bo) + x.
# The GEGLU output g and the residual stream h are written side by side into one (rows, 5C) buffer by their producers, so the
# concatenated operand is a plain row view: the (rows, C) intermediate is neither written nor re-read, one launch less per
# transformer / motion module (attention.py:154-161,316-320, motion_module.py:215-227,322-334).
FUSE_FF_TAIL = True
# The per-frame GroupNorm in front of proj_in (attention.py:124,135-146; motion_module.py:147-151) has no activation behind it:
# from this many pixels per frame on its scale / shift are folded into per-frame copies of the proj_in weights
# (ops.group_norm_fold_linear: B*F x C x C elements) and the GEMM reads the RAW rows - the apply pass (one read + one write of the
# activation) disappears: 81 -> 59 us per norm + proj_in at the 64x64 level, -0.25 ms per step.  Below it the weight copies
# cost what the pass they replace costs (32x32: 58 vs 58 us; 16x16: 64 vs 46) - tools/bench/gn_fold_bench.py.  0 = off.
GN_FOLD_MIN_HW = 4096
# GroupNorm + SiLU in front of a resnet's 3x3 convs (resnet.py:180-183,191-196) applied INSIDE the conv: the halo-reuse kernel
# normalises each halo chunk in LDS right after its direct-to-LDS load (ops.conv3x3(gn=...)), so the apply pass - one read and one
# write of the activation per conv - disappears; what remains of the norm is its read-only statistics pass.  Served where the conv
# runs on the halo kernel (W % 16 == 0); frames smaller than this many pixels keep the one-launch GroupNorm kernel.  0 = off.
GN_CONV_MIN_HW = 0
# attn1's q | k | v projection (orig_attention.py:598-600) as one LayerNorm-folded GEMM whose V columns are stored transposed by the same
# launch (emo_gemm_params.vt) instead of a q | k launch and a V^T launch that both read the rows.
MERGE_QKV = True


def ff_tail_weights(w_out, b_out, w2, b2):
    """ff.net.2 followed by proj_out as ONE linear map over the side-by-side operand [g | h] (FUSE_FF_TAIL):
    (g W2^T + b2 + h) Wo^T + bo = [g | h] [Wo W2 | Wo]^T + (Wo b2 + bo).  f32 in, f32 out ((C, 5C) weight, (C,) bias); proj_out may
    be a 1x1 conv (C, C, 1, 1) or a Linear (attention.py:135-146,154-161)."""
    wo = w_out.float().reshape(w_out.shape[0], -1)
    return torch.cat([wo @ w2.float(), wo], 1), wo @ b2.float() + b_out.float()


class _Ctx:
    """Per-forward geometry + reference-attention state."""

    def __init__(self, B, F, H, W):
        self.B, self.F, self.H, self.W = B, F, H, W
        self.bank_mode = None       # None | 'write' | 'read'
        self.banks = {}             # prefix -> rows tensor (B_ref*L, C) in compute dtype
        self.bank_rows = 0          # B_ref
        self.written = {}
        self.uc_batches = 0         # number of leading (b f) batches that skip the bank
        self.bank_skip = 0          # leading bank rows that were not materialised (cond-only ReferenceNet pass under CFG)
        self.bank_kv = {}           # prefix -> (K rows, V^T, L): bank projections precomputed by the sampler
        self.bank_row = None        # device int32 word: the row of bank_kv in use
        self.active = ()


class _State:
    """activations handed between the three forward stages"""


_STOP = object()   # returned by the gutted last transformer of the ReferenceNet: the rest of the pass is dead


class UNet3DConditionModel:
    def __init__(self, **kwargs):
        self._has_out = kwargs.pop("_has_out", True)
        self._controlnet = kwargs.pop("_controlnet", None)   # ControlNetModel: conditioning-embedding channels
        self._gut_last = kwargs.pop("_gut_last_transformer", False)   # AppearanceEncoderModel (appearance_encoder.py:613-621)
        self.config: FrozenConfig = normalize_unet_config(kwargs)
        self.spec: UNetSpec = build_spec(self.config, has_out=self._has_out, controlnet=self._controlnet,
                                         gut_last_transformer=self._gut_last)
        self.sample_size = self.config["sample_size"]
        self.in_channels = self.config["in_channels"]
        self.num_upsamplers = len(self.config["block_out_channels"]) - 1
        self.dtype = torch.float32
        self.device = torch.device("cpu")
        self._shapes = param_shapes(self.spec)
        self._master = None   # reference-layout fp32 tensors (device)
        self._w = None        # packed tensors
        self._reference_control = None

    # ------------------------------------------------------------------ torch-module-like surface
    def eval(self):
        return self

    def requires_grad_(self, *_):
        return self

    def parameters(self):
        return iter(self._master.values()) if self._master else iter(())

    def state_dict(self):
        if self._master is None:
            raise EmoHipError("no weights loaded")
        return {k: v.detach().cpu() for k, v in self._master.items()}

    def load_state_dict(self, sd, strict=True):
        """torch semantics: returns (missing, unexpected) relative to the INCOMING dict; strict raises on either.  Partial
        loads accumulate (the reference workflow is from_pretrained_2d(strict=False) followed by a second strict=False load of
        the motion-module checkpoint, animation.py:116-135); the model (re-)packs whenever the merged master is complete."""
        missing = [k for k in self._shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: {len(missing)} missing key(s) {missing[:5]}, "
                               f"{len(unexpected)} unexpected key(s) {unexpected[:5]}")
        for k, shp in self._shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shp)}")
        master = dict(self._master or {})
        for k in self._shapes:
            if k in sd:
                master[k] = sd[k].detach().to(torch.float32)
        self._master = master
        self._w = None                      # packed weights are stale from here on
        if not self._absent_keys():
            self._pack()
        return missing, unexpected

    def _absent_keys(self):
        return [k for k in self._shapes if self._master is None or k not in self._master]

    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            else:
                device = torch.device(a)
        if dtype is not None:
            if dtype not in (torch.float32, torch.bfloat16, torch.float16):
                raise EmoHipError(f"compute dtype {dtype} not supported (float32 | bfloat16 | float16)")
            self.dtype = dtype
        if device is not None:
            self.device = torch.device(device)
        if self._master is not None and not self._absent_keys():
            self._pack()
        return self

    def cuda(self):
        return self.to("cuda")

    @classmethod
    def from_config(cls, config, **extra):
        from .config import UNET_DEFAULTS
        kw = {k: v for k, v in dict(config).items() if k in UNET_DEFAULTS}
        kw.update({k: v for k, v in extra.items() if k in UNET_DEFAULTS})
        return cls(**kw)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """unet_controlnet.py:485-525: config.json + diffusion_pytorch_model.bin, 2D->3D block names,
        strict=False load, prints missing/unexpected counts."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file) as f:
            config = json.load(f)
        config["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
        config["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
        model = cls.from_config(config, **(unet_additional_kwargs or {}))
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        state_dict = torch.load(model_file, map_location="cpu")
        m, u = model.load_state_dict(state_dict, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        return model

    # ------------------------------------------------------------------ weight packing
    def _pack(self):
        if self.device.type != "cuda":
            return  # packed on the first .to('cuda'); the product path has no CPU execution
        absent = self._absent_keys()
        if absent:
            raise EmoHipError(f"UNet3DConditionModel: {len(absent)} parameter(s) were never loaded, e.g. {absent[:5]} "
                              "(load_state_dict(strict=False) leaves the model unusable until every key has a value)")
        dev, dtp = self.device, self.dtype
        m = {k: v.to(dev) for k, v in self._master.items()}
        self._master = m
        w = {}
        V = ops.vec(dtp)

        def lin(key):  # [N][K]
            t = m[key]
            return t.reshape(t.shape[0], -1).to(dtp).contiguous()

        def f32(key):
            return m[key].float().contiguous()

        def conv3(key):  # OIHW -> [Cout][ky][kx][Cin_pad]
            t = m[key]
            co, ci = t.shape[0], t.shape[1]
            cip = _round_up(ci, 8)
            o = torch.zeros(co, 3, 3, cip, device=dev, dtype=torch.float32)
            o[..., :ci] = t.permute(0, 2, 3, 1)
            return o.reshape(co, 9 * cip).to(dtp).contiguous()

        def ln_fold(wt, b, norm):
            """(W, bias) of a Linear behind LayerNorm `norm` -> (W * gamma rounded to the compute dtype, its row sums, bias +
            W . beta): LN(x) W^T + b = rstd (x W'^T - mean colsum) + b'."""
            g_, be = m[norm + ".weight"].float(), m[norm + ".bias"].float()
            wt = wt.reshape(wt.shape[0], -1).float()
            wp = (wt * g_[None, :]).to(dtp)
            bp = wt.to(dtp).float() @ be
            if b is not None:
                bp = bp + b.float()
            return wp.contiguous(), wp.float().sum(1).contiguous(), bp.contiguous()

        def geglu_rows(t):  # interleave (32 value, 32 gate) rows / entries
            n = t.shape[0] // 2
            if n % 32:
                raise EmoHipError("GEGLU width must be a multiple of 32")
            return torch.cat([t[:n].reshape(n // 32, 32, *t.shape[1:]), t[n:].reshape(n // 32, 32, *t.shape[1:])], 1).reshape(t.shape)

        def geglu_ln(prefix, norm):
            wp, cs, bp = ln_fold(m[prefix + ".weight"], m[prefix + ".bias"], norm)
            return geglu_rows(wp).contiguous(), geglu_rows(cs).contiguous(), geglu_rows(bp).contiguous()

        fold = self._fold_ln = bool(FOLD_LAYERNORM)
        self._fuse_tail = bool(FUSE_FF_TAIL)

        def ff_tail(ff2, proj_out):
            wt, bt = ff_tail_weights(m[proj_out + ".weight"], m[proj_out + ".bias"], m[ff2 + ".weight"], m[ff2 + ".bias"])
            return wt.to(dtp).contiguous(), bt.contiguous()

        def geglu(prefix):  # interleave (32 value, 32 gate)
            wt, b = m[prefix + ".weight"], m[prefix + ".bias"]
            n = wt.shape[0] // 2
            if n % 32:
                raise EmoHipError("GEGLU width must be a multiple of 32")
            wv, wg = wt[:n].reshape(n // 32, 32, -1), wt[n:].reshape(n // 32, 32, -1)
            bv, bg = b[:n].reshape(n // 32, 32), b[n:].reshape(n // 32, 32)
            return (torch.cat([wv, wg], 1).reshape(2 * n, -1).to(dtp).contiguous(),
                    torch.cat([bv, bg], 1).reshape(2 * n).float().contiguous())

        spec = self.spec
        w["conv_in.w"], w["conv_in.b"] = conv3("conv_in.weight"), f32("conv_in.bias")
        for n_ in ("linear_1", "linear_2"):
            w[f"time_embedding.{n_}.w"], w[f"time_embedding.{n_}.b"] = lin(f"time_embedding.{n_}.weight"), f32(f"time_embedding.{n_}.bias")
        if "class_embedding.weight" in m:                       # nn.Embedding(num_class_embeds, time_embed_dim) (unet_controlnet.py:120-121)
            w["class_embedding.table"] = m["class_embedding.weight"].to(dtp).contiguous()
        elif "class_embedding.linear_1.weight" in m:            # TimestepEmbedding (unet_controlnet.py:122-123)
            for n_ in ("linear_1", "linear_2"):
                w[f"class_embedding.{n_}.w"], w[f"class_embedding.{n_}.b"] = lin(f"class_embedding.{n_}.weight"), f32(f"class_embedding.{n_}.bias")
        half = spec.cfg["block_out_channels"][0] // 2
        expo = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - spec.cfg["freq_shift"])  # embeddings.py:47-51
        w["time_freqs"] = torch.exp(expo).to(dev)
        temb_w, temb_b, off = [], [], 0
        self._temb_off = {}
        for blk in spec.down + [spec.mid] + spec.up:
            for r in blk.resnets:
                p = r.prefix
                w[p + ".norm1.g"], w[p + ".norm1.b"] = f32(p + ".norm1.weight"), f32(p + ".norm1.bias")
                w[p + ".norm2.g"], w[p + ".norm2.b"] = f32(p + ".norm2.weight"), f32(p + ".norm2.bias")
                w[p + ".conv1.w"], w[p + ".conv1.b"] = conv3(p + ".conv1.weight"), f32(p + ".conv1.bias")
                w[p + ".conv2.w"], w[p + ".conv2.b"] = conv3(p + ".conv2.weight"), f32(p + ".conv2.bias")
                if r.has_shortcut:
                    w[p + ".sc.w"], w[p + ".sc.b"] = lin(p + ".conv_shortcut.weight"), f32(p + ".conv_shortcut.bias")
                temb_w.append(m[p + ".time_emb_proj.weight"])
                temb_b.append(m[p + ".time_emb_proj.bias"])
                self._temb_off[p] = off
                off += r.cout
            for a in blk.attentions:
                if a is None:
                    continue
                p = a.prefix
                tb = p + ".transformer_blocks.0"
                w[p + ".norm.g"], w[p + ".norm.b"] = f32(p + ".norm.weight"), f32(p + ".norm.bias")
                w[p + ".proj_in.w"], w[p + ".proj_in.b"] = lin(p + ".proj_in.weight"), f32(p + ".proj_in.bias")
                if a.gutted:   # ReferenceNet's last transformer: norm, proj_in, norm1 only
                    w[f"{tb}.norm1.g"], w[f"{tb}.norm1.b"] = f32(f"{tb}.norm1.weight"), f32(f"{tb}.norm1.bias")
                    continue
                w[p + ".proj_out.w"], w[p + ".proj_out.b"] = lin(p + ".proj_out.weight"), f32(p + ".proj_out.bias")
                for n_ in ("norm1", "norm2", "norm3"):
                    w[f"{tb}.{n_}.g"], w[f"{tb}.{n_}.b"] = f32(f"{tb}.{n_}.weight"), f32(f"{tb}.{n_}.bias")
                w[tb + ".attn1.qk"] = torch.cat([m[tb + ".attn1.to_q.weight"], m[tb + ".attn1.to_k.weight"]], 0).to(dtp).contiguous()
                w[tb + ".attn1.k"] = lin(tb + ".attn1.to_k.weight")
                w[tb + ".attn1.v"] = lin(tb + ".attn1.to_v.weight")
                w[tb + ".attn1.o.w"], w[tb + ".attn1.o.b"] = lin(tb + ".attn1.to_out.0.weight"), f32(tb + ".attn1.to_out.0.bias")
                w[tb + ".attn2.k"] = lin(tb + ".attn2.to_k.weight")
                w[tb + ".attn2.v"] = lin(tb + ".attn2.to_v.weight")
                w[tb + ".attn2.o.w"], w[tb + ".attn2.o.b"] = lin(tb + ".attn2.to_out.0.weight"), f32(tb + ".attn2.to_out.0.bias")
                w[tb + ".ff2.w"], w[tb + ".ff2.b"] = lin(tb + ".ff.net.2.weight"), f32(tb + ".ff.net.2.bias")
                if self._fuse_tail:
                    w[tb + ".tail.w"], w[tb + ".tail.b"] = ff_tail(tb + ".ff.net.2", p + ".proj_out")
                if fold:
                    w[tb + ".attn1.qk_ln"] = ln_fold(torch.cat([m[tb + ".attn1.to_q.weight"], m[tb + ".attn1.to_k.weight"]], 0), None, tb + ".norm1")
                    w[tb + ".attn1.v_ln"] = ln_fold(m[tb + ".attn1.to_v.weight"], None, tb + ".norm1")
                    # q | k | v as ONE projection (the V columns stored transposed by the same launch: ops.gemm(vt_cols=)); the two
                    # launches above remain for geometries the split store does not serve
                    w[tb + ".attn1.qkv_ln"] = tuple(torch.cat([x_, y_], 0).contiguous() for x_, y_ in zip(w[tb + ".attn1.qk_ln"], w[tb + ".attn1.v_ln"]))
                    w[tb + ".attn2.q_ln"] = ln_fold(m[tb + ".attn2.to_q.weight"], None, tb + ".norm2")
                    w[tb + ".ff1_ln"] = geglu_ln(tb + ".ff.net.0.proj", tb + ".norm3")
                else:
                    w[tb + ".attn2.q"] = lin(tb + ".attn2.to_q.weight")
                    w[tb + ".ff1.w"], w[tb + ".ff1.b"] = geglu(tb + ".ff.net.0.proj")
            for mo in list(blk.motions) + [a.tam for a in blk.attentions if a is not None and a.tam is not None]:
                if mo is None:
                    continue
                p = mo.prefix + ".temporal_transformer"
                w[p + ".norm.g"], w[p + ".norm.b"] = f32(p + ".norm.weight"), f32(p + ".norm.bias")
                w[p + ".proj_in.w"], w[p + ".proj_in.b"] = lin(p + ".proj_in.weight"), f32(p + ".proj_in.bias")
                w[p + ".proj_out.w"], w[p + ".proj_out.b"] = lin(p + ".proj_out.weight"), f32(p + ".proj_out.bias")
                for bi in range(mo.n_blocks):
                    tb = f"{p}.transformer_blocks.{bi}"
                    for k in range(mo.n_attn):
                        ab = f"{tb}.attention_blocks.{k}"
                        wqkv = torch.cat([m[ab + ".to_q.weight"], m[ab + ".to_k.weight"], m[ab + ".to_v.weight"]], 0)
                        w[ab + ".o.w"], w[ab + ".o.b"] = lin(ab + ".to_out.0.weight"), f32(ab + ".to_out.0.bias")
                        if fold:
                            w[ab + ".qkv_ln"] = ln_fold(wqkv, None, f"{tb}.norms.{k}")
                            if mo.pe_len:   # (LN(x) + pe[f]) W^T = LN(x) W^T + (pe W^T)[f]: a per-frame row bias (motion_module.py:246-248,282-283)
                                w[ab + ".pe_w"] = (m[ab + ".pos_encoder.pe"][0].float() @ wqkv.to(dtp).float().t()).contiguous()
                        else:
                            w[ab + ".qkv"] = wqkv.to(dtp).contiguous()
                            w[f"{tb}.norms.{k}.g"], w[f"{tb}.norms.{k}.b"] = f32(f"{tb}.norms.{k}.weight"), f32(f"{tb}.norms.{k}.bias")
                            if mo.pe_len:
                                w[ab + ".pe"] = m[ab + ".pos_encoder.pe"][0].float().contiguous()
                    if fold:
                        w[tb + ".ff1_ln"] = geglu_ln(tb + ".ff.net.0.proj", tb + ".ff_norm")
                    else:
                        w[tb + ".ff_norm.g"], w[tb + ".ff_norm.b"] = f32(tb + ".ff_norm.weight"), f32(tb + ".ff_norm.bias")
                        w[tb + ".ff1.w"], w[tb + ".ff1.b"] = geglu(tb + ".ff.net.0.proj")
                    w[tb + ".ff2.w"], w[tb + ".ff2.b"] = lin(tb + ".ff.net.2.weight"), f32(tb + ".ff.net.2.bias")
                    if self._fuse_tail and bi == mo.n_blocks - 1:      # the LAST block's ff.net.2 composes with proj_out
                        w[tb + ".tail.w"], w[tb + ".tail.b"] = ff_tail(tb + ".ff.net.2", p + ".proj_out")
            for a in blk.attentions:      # VideoNet: the SpatialAttentionModule in front of each transformer (models/videonet.py:15-77)
                if a is None or a.sam is None:
                    continue
                sp = a.sam
                w[sp + ".norm_in.g"], w[sp + ".norm_in.b"] = f32(sp + ".norm_in.weight"), f32(sp + ".norm_in.bias")
                for n_ in ("proj_in", "to_q", "ffn", "proj_out"):
                    w[f"{sp}.{n_}.w"], w[f"{sp}.{n_}.b"] = lin(f"{sp}.{n_}.weight"), f32(f"{sp}.{n_}.bias")
                w[sp + ".k.w"], w[sp + ".k.b"] = lin(sp + ".to_k.weight"), f32(sp + ".to_k.bias")
                w[sp + ".v.w"], w[sp + ".v.b"] = lin(sp + ".to_v.weight"), f32(sp + ".to_v.bias")
                for n_ in ("norm1", "norm2"):
                    w[f"{sp}.{n_}.g"], w[f"{sp}.{n_}.b"] = f32(f"{sp}.{n_}.weight"), f32(f"{sp}.{n_}.bias")
            if blk.sampler:
                w[blk.sampler + ".w"], w[blk.sampler + ".b"] = conv3(blk.sampler + ".conv.weight"), f32(blk.sampler + ".conv.bias")
        w["temb_all.w"] = torch.cat(temb_w, 0).to(dtp).contiguous()
        w["temb_all.b"] = torch.cat(temb_b, 0).float().contiguous()
        if spec.has_out:
            w["conv_norm_out.g"], w["conv_norm_out.b"] = f32("conv_norm_out.weight"), f32("conv_norm_out.bias")
            w["conv_out.w"], w["conv_out.b"] = conv3("conv_out.weight"), f32("conv_out.bias")
        if spec.controlnet is not None:
            ce = "controlnet_cond_embedding"
            names = ["conv_in"] + [f"blocks.{i}" for i in range(2 * (len(spec.controlnet) - 1))] + ["conv_out"]
            for n_ in names:
                w[f"{ce}.{n_}.w"], w[f"{ce}.{n_}.b"] = conv3(f"{ce}.{n_}.weight"), f32(f"{ce}.{n_}.bias")
            from .spec import skip_channels
            for k in range(len(skip_channels(spec))):
                w[f"controlnet_down_blocks.{k}.w"], w[f"controlnet_down_blocks.{k}.b"] = lin(f"controlnet_down_blocks.{k}.weight"), f32(f"controlnet_down_blocks.{k}.bias")
            w["controlnet_mid_block.w"], w["controlnet_mid_block.b"] = lin("controlnet_mid_block.weight"), f32("controlnet_mid_block.bias")
        self._w = w
        self._pe_rows = {}

    # ------------------------------------------------------------------ blocks (all HIP launches)
    def _resnet(self, r, x, temb_all, c: _Ctx, H, W, scale=1.0, out=None):
        """resnet.py:177-207.  GN statistics are JOINT over the F frames of a batch row (5-D GroupNorm)."""
        w, p = self._w, r.prefix
        G, eps = self.config["norm_num_groups"], self.config["norm_eps"]
        n_img = c.B * c.F
        off = self._temb_off[p]
        temb = temb_all[:, off:off + r.cout]
        h = self._norm_act_conv(x, p + ".norm1", p + ".conv1", c, H, W, G, eps, rowbias=temb, rows_per_batch=c.F * H * W)
        sc = ops.gemm(x, w[p + ".sc.w"], w[p + ".sc.b"]) if r.has_shortcut else x
        return self._norm_act_conv(h, p + ".norm2", p + ".conv2", c, H, W, G, eps, inplace=True, residual=sc, out_scale=1.0 / scale, out=out)

    def _norm_act_conv(self, x, norm, conv, c: _Ctx, H, W, G, eps, inplace=False, **kw):
        """GroupNorm (joint over the F frames of a batch row) -> SiLU -> 3x3 conv (resnet.py:180-183,191-196; unet_controlnet.py:476-477)
        with the normalisation inside the conv where the halo-reuse kernel serves it (GN_CONV_MIN_HW)."""
        w = self._w
        n_img = c.B * c.F
        if GN_CONV_MIN_HW and H * W >= GN_CONV_MIN_HW and ops.conv_gn_fusable(x, w[conv + ".w"], n_img, H, W, kw.get("rowbias"), kw.get("rows_per_batch", 0)):
            coef = ops.group_norm_coeffs(x, w[norm + ".g"], w[norm + ".b"], c.B, G, eps)
            return ops.conv3x3(x, w[conv + ".w"], w[conv + ".b"], n_img, H, W, gn=(coef, c.F, True), **kw)[0]
        h = ops.group_norm(x, w[norm + ".g"], w[norm + ".b"], c.B, G, eps, True, out=x if inplace else None)
        return ops.conv3x3(h, w[conv + ".w"], w[conv + ".b"], n_img, H, W, **kw)[0]

    def _kv(self, rows, wk, wv, L):
        """K rows and V^T (per batch of L rows) projections of `rows` (nb*L, Cin)."""
        k = ops.gemm(rows, wk)
        vt = ops.gemm(rows, wv, transpose_rows=L, transpose_ld=_round_up(L, 8))
        return k, vt

    def transformer_prefixes(self):
        return [a.prefix for blk in self.spec.down + [self.spec.mid] + self.spec.up for a in blk.attentions
                if a is not None and not a.gutted]

    def context_kv(self, ctx, dtype=None):
        """attn2 K / V^T projections of a context (nb, L, D) for every transformer block: they depend on the context only
        (orig_attention.py:594-600), not on the timestep or the latents, so a sampling loop computes them once per clip and
        hands them to every step (`_ctx_kv=`).  Returns {prefix: (K rows, V^T)}."""
        if self._w is None:
            raise EmoHipError("context_kv: weights not loaded / model not on a HIP device")
        ctx = ctx.to(self.device)
        rows = ops.convert(ctx.float().reshape(-1, ctx.shape[2]), self.dtype)
        out = {}
        for p in self.transformer_prefixes():
            tb = p + ".transformer_blocks.0"
            out[p] = self._kv(rows, self._w[tb + ".attn2.k"], self._w[tb + ".attn2.v"], ctx.shape[1])
        return out

    def bank_kv(self, prefix, bank_rows, L):
        """attn1 K / V^T projections of reference-bank rows (n*L, C) with THIS model's to_k / to_v
        (mutual_self_attention.py:238-241: the bank is concatenated to the K/V input of attn1)."""
        tb = prefix + ".transformer_blocks.0"
        return self._kv(bank_rows, self._w[tb + ".attn1.k"], self._w[tb + ".attn1.v"], L)

    @staticmethod
    def _dup_rows(xh):
        """[x; x]: the two halves of a classifier-free-guidance batch after their shared prefix (two strided row copies)."""
        y = torch.empty(2 * xh.shape[0], xh.shape[1], device=xh.device, dtype=xh.dtype)
        ops.copy_cols(xh, y[:xh.shape[0]], 0)
        ops.copy_cols(xh, y[xh.shape[0]:], 0)
        return y

    def _transformer(self, a, x, ctx_rows, ctx_len, ctx_div, c: _Ctx, H, W, out=None, ctx_kv=None, shared_half=False):
        """attention.py:112-161 + 276-320, and the write/read hooks of mutual_self_attention.py:199-284.
        shared_half: x holds the FIRST half of the batch only and the second half is identical (a [uncond, cond] batch before its
        first cross-attention): GroupNorm, proj_in, LN1, q|k / V^T, self-attention and its output projection run on those rows,
        the result is duplicated in front of the cross-attention (c describes the FULL batch)."""
        w, p = self._w, a.prefix
        tb = p + ".transformer_blocks.0"
        C_, heads = a.channels, a.heads
        d = C_ // heads
        HW, nb = H * W, c.B * c.F
        scale = d ** -0.5
        nb_full = nb
        if shared_half:
            nb = nb // 2
        h = self._norm_proj_in(x, p, nb, HW, self.config["norm_num_groups"])
        # --- self attention (+ reference bank)
        fold = self._fold_ln and not a.gutted and not (c.bank_mode == "write" and p in c.active)
        LN_EPS = 1e-5   # nn.LayerNorm default (attention.py:240-252)
        if fold:        # LN1 folded into the q|k and V^T projections
            st = ops.layer_norm_stats(h, LN_EPS)
            both = None
            if MERGE_QKV:   # one launch: h is read once, q | k row-major, V^T by the V waves of the same tiles
                wq, cs, bq = w[tb + ".attn1.qkv_ln"]
                both = ops.gemm(h, wq, bq, ln=(cs, st), vt_cols=C_, vt_rows=HW, vt_ld=_round_up(HW, 8))
            if both is not None:
                qk, vt = both
            else:
                wq, cs, bq = w[tb + ".attn1.qk_ln"]
                qk = ops.gemm(h, wq, bq, ln=(cs, st))
                wv, cs, bv = w[tb + ".attn1.v_ln"]
                vt = ops.gemm(h, wv, bv, ln=(cs, st), transpose_rows=HW, transpose_ld=_round_up(HW, 8))
        else:
            n1 = ops.layer_norm(h, w[tb + ".norm1.g"], w[tb + ".norm1.b"])
            if c.bank_mode == "write" and p in c.active:
                c.written[p] = n1  # LN1 output (mutual_self_attention.py:230)
            if a.gutted:
                # ReferenceNet's last transformer (appearance_encoder.py:613-621): behind LN1 there are no parameters and the
                # model's output is discarded - the pass ends here
                return _STOP
            qk = ops.gemm(n1, w[tb + ".attn1.qk"])
            vt = ops.gemm(n1, w[tb + ".attn1.v"], transpose_rows=HW, transpose_ld=_round_up(HW, 8))
        kw = {}
        if c.bank_mode == "read" and p in c.active and (p in c.banks or p in c.bank_kv):
            if p in c.bank_kv:    # projected once per group of timesteps by the sampler (pipeline._reference_group)
                kb, vbt, Lb = c.bank_kv[p]
                kw = dict(k1=kb, v1t=vbt, Lk1=Lb, seg1_div=nb, seg1_first_batch=c.uc_batches, seg1_row=c.bank_row)
            else:
                bank = c.banks[p]
                Lb = bank.shape[0] // c.bank_rows
                kb, vbt = self._kv(bank, w[tb + ".attn1.k"], w[tb + ".attn1.v"], Lb)
                kw = dict(k1=kb, v1t=vbt, Lk1=Lb, seg1_div=c.F, seg1_first_batch=c.uc_batches, seg1_skip=c.bank_skip)
        att = ops.attention(qk[:, :C_], qk[:, C_:], vt, HW, B=nb, Lq=HW, heads=heads, d=d, scale=scale, **kw)
        h = ops.gemm(att, w[tb + ".attn1.o.w"], w[tb + ".attn1.o.b"], residual=h)
        if shared_half:     # the halves part ways at the text / audio cross-attention
            h, x, nb = self._dup_rows(h), self._dup_rows(x), nb_full
        # --- cross attention to the text / audio context
        if self._fold_ln:
            wq, cs, bq = w[tb + ".attn2.q_ln"]
            q2 = ops.gemm(h, wq, bq, ln=(cs, ops.layer_norm_stats(h, LN_EPS)))
        else:
            n2 = ops.layer_norm(h, w[tb + ".norm2.g"], w[tb + ".norm2.b"])
            q2 = ops.gemm(n2, w[tb + ".attn2.q"])
        if ctx_kv is not None:
            kc, vct = ctx_kv[p]
        else:
            kc, vct = self._kv(ctx_rows, w[tb + ".attn2.k"], w[tb + ".attn2.v"], ctx_len)
        att = ops.attention(q2, kc, vct, ctx_len, B=nb, Lq=HW, heads=heads, d=d, scale=scale, seg0_div=ctx_div)
        gh, g_out, h_out = self._tail_buffer(h, C_)
        h = ops.gemm(att, w[tb + ".attn2.o.w"], w[tb + ".attn2.o.b"], residual=h, out=h_out)
        # --- GEGLU feed-forward
        if self._fold_ln:
            wf, cs, bf = w[tb + ".ff1_ln"]
            g = ops.gemm(h, wf, bf, geglu=True, ln=(cs, ops.layer_norm_stats(h, LN_EPS)), out=g_out)
        else:
            n3 = ops.layer_norm(h, w[tb + ".norm3.g"], w[tb + ".norm3.b"])
            g = ops.gemm(n3, w[tb + ".ff1.w"], w[tb + ".ff1.b"], geglu=True, out=g_out)
        if gh is not None:      # ff.net.2 + residual + proj_out + residual in one pass over [g | h]
            return ops.gemm(gh, w[tb + ".tail.w"], w[tb + ".tail.b"], residual=x, out=out)
        h = ops.gemm(g, w[tb + ".ff2.w"], w[tb + ".ff2.b"], residual=h)
        return ops.gemm(h, w[p + ".proj_out.w"], w[p + ".proj_out.b"], residual=x, out=out)

    def _norm_proj_in(self, x, p, nb, HW, groups):
        """GroupNorm(eps 1e-6, per frame) + proj_in; folded into per-frame weights where a frame is large (GN_FOLD_MIN_HW)."""
        w = self._w
        if GN_FOLD_MIN_HW and HW >= GN_FOLD_MIN_HW and HW % 256 == 0:
            wn, rb = ops.group_norm_fold_linear(x, w[p + ".norm.g"], w[p + ".norm.b"], nb, groups, 1e-6, w[p + ".proj_in.w"], w[p + ".proj_in.b"])
            return ops.gemm(x, wn, rb, w_slab_rows=HW)
        h = ops.group_norm(x, w[p + ".norm.g"], w[p + ".norm.b"], nb, groups, 1e-6, False)
        return ops.gemm(h, w[p + ".proj_in.w"], w[p + ".proj_in.b"])

    def _tail_buffer(self, h, C_):
        """(rows, 5C) buffer [g | h] of the fused ff.net.2 + proj_out GEMM and its two column views (FUSE_FF_TAIL); Nones when off."""
        if not self._fuse_tail:
            return None, None, None
        gh = torch.empty(h.shape[0], 5 * C_, device=h.device, dtype=h.dtype)
        return gh, gh[:, :4 * C_], gh[:, 4 * C_:]

    def _motion(self, mo, x, c: _Ctx, H, W, out=None):
        """motion_module.py:139-163,215-227,275-334 (VanillaTemporalModule)."""
        w = self._w
        p = mo.prefix + ".temporal_transformer"
        tb = p + ".transformer_blocks.0"
        C_, heads = mo.channels, mo.heads
        d = C_ // heads
        HW, nb = H * W, c.B * c.F
        if mo.pe_len and c.F > mo.pe_len:
            raise ValueError(f"video_length {c.F} exceeds temporal_position_encoding_max_len {mo.pe_len}")
        h = self._norm_proj_in(x, p, nb, HW, 32)  # norm_num_groups=32 default (:104)
        gh = None
        for bi in range(mo.n_blocks):       # TemporalTransformer3DModel.transformer_blocks (motion_module.py:139-163; 1 in configs/inference.yaml)
            tb = f"{p}.transformer_blocks.{bi}"
            last_block = bi == mo.n_blocks - 1
            g_out = h_out = None
            for k in range(mo.n_attn):
                ab = f"{tb}.attention_blocks.{k}"
                if self._fold_ln:   # LN folded into the q|k|v projection, the positional encoding into a per-frame row bias
                    wq, cs, bq = w[ab + ".qkv_ln"]
                    kw = {}
                    if mo.pe_len:
                        key = (ab, c.B, c.F)
                        if key not in self._pe_rows:
                            self._pe_rows[key] = w[ab + ".pe_w"][:c.F].repeat(c.B, 1).contiguous()
                        kw = dict(rowbias=self._pe_rows[key], rows_per_batch=HW)
                    qkv = ops.gemm(h, wq, bq, ln=(cs, ops.layer_norm_stats(h, 1e-5)), **kw)
                else:
                    n = ops.layer_norm(h, w[f"{tb}.norms.{k}.g"], w[f"{tb}.norms.{k}.b"], pe=w.get(ab + ".pe"), rows_per_frame=HW, frames=c.F)
                    qkv = ops.gemm(n, w[ab + ".qkv"])
                att = ops.temporal_attention(qkv, c.B, c.F, HW, heads, d, d ** -0.5)
                if last_block and k == mo.n_attn - 1:
                    gh, g_out, h_out = self._tail_buffer(h, C_)
                h = ops.gemm(att, w[ab + ".o.w"], w[ab + ".o.b"], residual=h, out=h_out)
            if self._fold_ln:
                wf, cs, bf = w[tb + ".ff1_ln"]
                g = ops.gemm(h, wf, bf, geglu=True, ln=(cs, ops.layer_norm_stats(h, 1e-5)), out=g_out)
            else:
                n = ops.layer_norm(h, w[tb + ".ff_norm.g"], w[tb + ".ff_norm.b"])
                g = ops.gemm(n, w[tb + ".ff1.w"], w[tb + ".ff1.b"], geglu=True, out=g_out)
            if last_block and gh is not None:
                return ops.gemm(gh, w[tb + ".tail.w"], w[tb + ".tail.b"], residual=x, out=out)
            h = ops.gemm(g, w[tb + ".ff2.w"], w[tb + ".ff2.b"], residual=h)
        return ops.gemm(h, w[p + ".proj_out.w"], w[p + ".proj_out.b"], residual=x, out=out)

    # ------------------------------------------------------------------ forward (three stages so that the sampler can
    # overlap the ReferenceNet pass with the bank-independent down path on a second HIP stream)
    def _begin(self, sample, timestep, encoder_hidden_states, audio_features=None, speed_embeddings=None,
               down_block_additional_residuals=None, mid_block_additional_residual=None, add_after_conv_in=None, _ctx_kv=None,
               halves_identical=False, class_labels=None):
        if self._w is None:
            absent = self._absent_keys()
            raise EmoHipError("UNet3DConditionModel: weights not loaded / model not on a HIP device "
                              "(load_state_dict + .to('cuda')); there is no CPU execution path" +
                              (f"; {len(absent)} parameter(s) have no value yet, e.g. {absent[:3]}" if absent and self._master else ""))
        if sample.dim() != 5:
            raise ValueError(f"Expected sample to have ndim=5, but got ndim={sample.dim()}.")
        cfg, w, dtp = self.config, self._w, self.dtype
        B, Cin, F, H, W = sample.shape
        if Cin != cfg["in_channels"]:
            raise ValueError(f"sample has {Cin} channels, expected {cfg['in_channels']}")
        up = 2 ** self.num_upsamplers
        # unet_controlnet.py:357-365: inputs that are not a multiple of 2^num_upsamplers forward the skip's size to the upsamplers
        forward_upsample_size = bool(H % up or W % up)
        dev = self.device
        s = _State()
        s.c = _Ctx(B, F, H, W)
        s.B, s.F, s.H, s.W = B, F, H, W
        s.forward_upsample_size, s.skip_hw = forward_upsample_size, []
        sample = sample.to(dev)
        # time (unet_controlnet.py:376-398)
        if not torch.is_tensor(timestep):
            timesteps = torch.tensor([timestep], dtype=torch.int64, device=dev)
        else:
            timesteps = timestep.reshape(-1).to(device=dev, dtype=torch.int64)
        timesteps = timesteps.expand(B).contiguous()
        t_emb = ops.timestep_embedding(timesteps, w["time_freqs"], cfg["block_out_channels"][0], cfg["flip_sin_to_cos"], dtp)
        e = ops.gemm(t_emb, w["time_embedding.linear_1.w"], w["time_embedding.linear_1.b"])
        emb = ops.gemm(ops.silu(e), w["time_embedding.linear_2.w"], w["time_embedding.linear_2.b"])
        has_class = cfg["class_embed_type"] is not None or cfg["num_class_embeds"] is not None
        if has_class:       # unet_controlnet.py:400-408
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            cl = class_labels.to(dev)
            if cfg["class_embed_type"] == "timestep":
                ce = ops.timestep_embedding(cl.reshape(-1).to(torch.int64).expand(B).contiguous(), w["time_freqs"], cfg["block_out_channels"][0],
                                            cfg["flip_sin_to_cos"], dtp)
                ce = ops.gemm(ops.silu(ops.gemm(ce, w["class_embedding.linear_1.w"], w["class_embedding.linear_1.b"])),
                              w["class_embedding.linear_2.w"], w["class_embedding.linear_2.b"])
            elif cfg["class_embed_type"] == "identity":
                ce = ops.convert(cl.float().reshape(B, -1).contiguous(), dtp)
            else:
                ce = ops.gather_rows(w["class_embedding.table"], cl.reshape(-1).to(torch.int32).expand(B).contiguous())
            emb = ops.add(emb, ce)
        if speed_embeddings is not None:  # EMO extension: class-embedding slot (unet_controlnet.py:400-408)
            se = ops.convert(speed_embeddings.to(dev).float().reshape(B, -1), dtp)
            emb = ops.add(emb, se)
        s.temb_all = ops.convert(ops.gemm(ops.silu(emb), w["temb_all.w"], w["temb_all.b"]), torch.float32)
        # context rows
        ctx = encoder_hidden_states if audio_features is None else audio_features
        ctx = ctx.to(dev)
        if ctx.shape[0] == B * F:
            s.ctx_div = 1
        elif ctx.shape[0] == B:
            s.ctx_div = F
        elif ctx.shape[0] == 1 and _ctx_kv is not None:
            s.ctx_div = B * F      # one context shared by every batch row (the sampler's batched ReferenceNet pass)
        else:
            raise ValueError(f"encoder_hidden_states batch {ctx.shape[0]} is neither B={B} nor B*F={B * F}")
        s.ctx_len = ctx.shape[1]
        s.ctx_kv = _ctx_kv        # {prefix: (K rows, V^T)} from context_kv(): attn2 projections hoisted out of the step
        s.ctx_rows = None if _ctx_kv is not None else ops.convert(ctx.float().reshape(-1, ctx.shape[2]), dtp)
        s.ctrl = (down_block_additional_residuals, mid_block_additional_residual)
        x = ops.ncfhw_to_rows(sample, dtp, cpad=_round_up(Cin, 8))
        if cfg["center_input_sample"]:      # sample = 2 * sample - 1.0 (unet_controlnet.py:371-373); the pad channels stay zero
            m1 = torch.zeros(1, x.shape[1], device=dev, dtype=dtp)
            m1[:, :Cin] = -1.0
            x = ops.add_rowbias(ops.add(x, x), m1, x.shape[0])
        # Zero-copy skip connections: every skip tensor is produced straight into the RIGHT columns of the buffer the up
        # path will read as cat([hidden, skip]) (unet_3d_blocks.py:627-629), and the up path's producers write `hidden`
        # into its LEFT columns - the concatenation never runs.  (ControlNet residuals re-materialise the skips: old path.)
        s.zero_copy = down_block_additional_residuals is None and len(self.spec.up) > 0
        s.skips, s.n_pushed = [], 0
        # SHARED PREFIX under classifier-free guidance (the sampler's [uncond, cond] batch, EMOAnimationPipeline.py:759-763): the two
        # halves of the batch carry the SAME latents, timestep and (speed) embedding and differ only from the first text
        # cross-attention on - conv_in, the first resnet and the first transformer up to and including its self-attention
        # output projection (no reference bank in the down path under fusion_blocks="midup") are computed ONCE on half the rows
        # and duplicated.  Same arithmetic, half the rows: ~0.8 ms of the 44 ms step (a 64x64-level self-attention alone is 0.45).
        s.dup = SHARE_CFG_PREFIX and bool(halves_identical) and B % 2 == 0 and add_after_conv_in is None and speed_embeddings is None and not has_class
        s.x_half = None
        if s.dup:
            Mh = (B // 2) * F * H * W
            s.x_half, _, _ = ops.conv3x3(x[:Mh], w["conv_in.w"], w["conv_in.b"], (B // 2) * F, H, W)
            s.x = self._skip_slot(s, B * F * H * W, self.spec.down[0].resnets[0].cin)
            if s.x is None:
                s.x = torch.empty(B * F * H * W, s.x_half.shape[1], device=dev, dtype=dtp)
            ops.copy_cols(s.x_half, s.x[:Mh], 0)
            ops.copy_cols(s.x_half, s.x[Mh:], 0)
        else:
            s.x, _, _ = ops.conv3x3(x, w["conv_in.w"], w["conv_in.b"], B * F, H, W, out=self._skip_slot(s, B * F * H * W, self.spec.down[0].resnets[0].cin))
        if add_after_conv_in is not None:   # ControlNet: sample += controlnet_cond_embedding(cond) (controlnet.py:523-525)
            s.x = ops.add(s.x, add_after_conv_in)
        self._push_skip(s, s.x, (H, W))
        s.h, s.w = H, W
        return s

    # ---- skip-connection plumbing
    def _up_resnets(self):
        return [r for blk in self.spec.up for r in blk.resnets]

    def _skip_slot(self, s, rows, c2):
        """View for the next skip tensor (c2 channels): the right columns of its future concat buffer."""
        if not s.zero_copy:
            return None
        ups = self._up_resnets()
        c1 = ups[len(ups) - 1 - s.n_pushed].cin - c2
        assert c1 > 0, (c1, c2)
        s.pending = torch.empty(rows, c1 + c2, device=self.device, dtype=self.dtype)
        return s.pending[:, c1:]

    def _push_skip(self, s, x, hw=None):
        s.skips.append(s.pending if s.zero_copy else x)
        s.skip_hw.append(hw)
        s.n_pushed += 1

    @staticmethod
    def _hidden_slot(s, skips, c1):
        """View for the up path's next `hidden` (c1 channels): the left columns of the buffer on top of the skip stack."""
        if not s.zero_copy or not skips:
            return None
        assert skips[-1].shape[1] > c1
        return skips[-1][:, :c1]

    def _run_down(self, s):
        w, spec, dtp, dev = self._w, self.spec, self.dtype, self.device
        x, c, h_, w_ = s.x, s.c, s.h, s.w
        first = True
        for blk in spec.down:
            for r, a, mo in zip(blk.resnets, blk.attentions, blk.motions):
                slot = self._skip_slot(s, x.shape[0], r.cout)          # the sub-block's last op writes the skip in place
                # the shared prefix of a [uncond, cond] batch (see _begin): first resnet + first transformer's self-attention
                # half on B / 2 rows; not when that transformer reads or writes a reference bank (fusion_blocks="full")
                dup = first and s.dup and a is not None and not a.gutted and not (c.bank_mode is not None and a.prefix in c.active)
                first = False
                if dup:
                    ch = _Ctx(c.B // 2, c.F, c.H, c.W)
                    xh = self._resnet(r, s.x_half, s.temb_all[:c.B // 2], ch, h_, w_)
                    x = self._transformer(a, xh, s.ctx_rows, s.ctx_len, s.ctx_div, c, h_, w_, out=slot if mo is None else None, ctx_kv=s.ctx_kv,
                                          shared_half=True)
                    if mo is not None:
                        x = self._motion(mo, x, c, h_, w_, out=slot)
                    self._push_skip(s, x, (h_, w_))
                    continue
                x = self._resnet(r, x, s.temb_all, c, h_, w_, out=slot if (a is None and mo is None) else None)
                if a is not None:
                    x = self._transformer(a, x, s.ctx_rows, s.ctx_len, s.ctx_div, c, h_, w_, out=slot if mo is None else None, ctx_kv=s.ctx_kv)
                if mo is not None:
                    x = self._motion(mo, x, c, h_, w_, out=slot)
                self._push_skip(s, x, (h_, w_))
            if blk.sampler:
                ho, wo = (h_ - 1) // 2 + 1, (w_ - 1) // 2 + 1          # 3x3, stride 2, padding 1
                slot = self._skip_slot(s, s.B * s.F * ho * wo, x.shape[1])
                x, h_, w_ = ops.conv3x3(x, w[blk.sampler + ".w"], w[blk.sampler + ".b"], s.B * s.F, h_, w_, stride=2, out=slot)
                self._push_skip(s, x, (h_, w_))
        down_res, mid_res = s.ctrl
        if down_res is not None and mid_res is not None:
            s.skips = [ops.add(sk, ops.ncfhw_to_rows(r.to(dev), dtp)) for sk, r in zip(s.skips, down_res)]  # unet_controlnet.py:430-439
        s.x, s.h, s.w = x, h_, w_

    def _run_mid(self, s, x, out=None):
        """mid block (unet_3d_blocks.py UNetMidBlock3DCrossAttn): resnet, transformer, [motion], resnet"""
        spec, c, h_, w_ = self.spec, s.c, s.h, s.w
        sc = self.config["mid_block_scale_factor"]
        x = self._resnet(spec.mid.resnets[0], x, s.temb_all, c, h_, w_, sc)
        x = self._transformer(spec.mid.attentions[0], x, s.ctx_rows, s.ctx_len, s.ctx_div, c, h_, w_, ctx_kv=s.ctx_kv)
        if spec.mid.motions[0] is not None:
            x = self._motion(spec.mid.motions[0], x, c, h_, w_)
        return self._resnet(spec.mid.resnets[1], x, s.temb_all, c, h_, w_, sc, out=out)

    def _run_rest(self, s, return_rows=False):
        cfg, w, spec, dtp, dev = self.config, self._w, self.spec, self.dtype, self.device
        x, c, h_, w_ = s.x, s.c, s.h, s.w
        B, F = s.B, s.F
        down_res, mid_res = s.ctrl
        skips = s.skips
        x = self._run_mid(s, x, out=self._hidden_slot(s, skips, spec.mid.resnets[1].cout))
        if down_res is not None and mid_res is not None:
            x = ops.add(x, ops.ncfhw_to_rows(mid_res.to(dev), dtp))
        for blk in spec.up:
            n_res = len(blk.resnets)
            for ri, (r, a, mo) in enumerate(zip(blk.resnets, blk.attentions, blk.motions)):
                if s.zero_copy:
                    x = skips.pop()                       # [hidden | skip] is already in place
                else:
                    x = ops.concat_cols(x, skips.pop())  # unet_3d_blocks.py:627-629
                # the op that produces the next `hidden` writes it into the next concat buffer - unless an upsampler follows
                last = ri == n_res - 1 and blk.sampler
                slot = None if last else self._hidden_slot(s, skips, r.cout)
                x = self._resnet(r, x, s.temb_all, c, h_, w_, out=slot if (a is None and mo is None) else None)
                if a is not None:
                    x = self._transformer(a, x, s.ctx_rows, s.ctx_len, s.ctx_div, c, h_, w_, out=slot if mo is None else None, ctx_kv=s.ctx_kv)
                    if x is _STOP:   # ReferenceNet: nothing behind the last bank write is ever read
                        break
                if mo is not None:
                    x = self._motion(mo, x, c, h_, w_, out=slot)
            if x is _STOP:
                break
            if blk.sampler:
                # unet_controlnet.py:456-459 / resnet.py:74-82: with forward_upsample_size the interpolation targets the size of
                # the next skip tensor instead of x2
                tgt = s.skip_hw[len(skips) - 1] if (s.forward_upsample_size and skips) else (2 * h_, 2 * w_)
                x, h_, w_ = ops.conv3x3(x, w[blk.sampler + ".w"], w[blk.sampler + ".b"], B * F, h_, w_, upsample_to=tgt,
                                        out=None if not s.zero_copy or not skips else self._hidden_slot(s, skips, x.shape[1]))
        rc = self._reference_control
        if rc is not None:
            rc._finish(c, self)
        if not spec.has_out or x is _STOP:
            return None
        x = self._norm_act_conv(x, "conv_norm_out", "conv_out", c, h_, w_, cfg["norm_num_groups"], cfg["norm_eps"])
        if return_rows:  # the sampler consumes NHWC rows directly (no layout round trip)
            return x
        return ops.rows_to_ncfhw(x, B, cfg["out_channels"], F, h_, w_)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
                mid_block_additional_residual: Optional[torch.Tensor] = None, return_dict: bool = True,
                audio_features=None, speed_embeddings=None, _return_rows=False, _ctx_kv=None,
                _halves_identical=False) -> Union[UNet3DConditionOutput, Tuple]:
        """unet_controlnet.py:328-483.  sample (B,C,F,h,w); timestep Tensor|int|float;
        encoder_hidden_states (B|B*F, L, D).  EMO extension kwargs (EMOAnimationPipeline.py:783-784):
        audio_features (B*F, L_a, D) per-frame attn2 context; speed_embeddings (B, 4*C0) added to emb."""
        # attention_mask: the reference prepares it (unet_controlnet.py:366-369) and hands it to its blocks, whose forwards never pass it
        # on to their transformers (unet_3d_blocks.py:276-283,384-410,618-660; Transformer3DModel.forward has no such parameter): a DEAD
        # input - the reference's output with a mask equals its output without (tests/golden/unet_switches.safetensors attention_mask/out).
        # Accepted and ignored, like upstream.
        if attention_mask is not None and not torch.is_tensor(attention_mask):
            raise TypeError("attention_mask must be a tensor or None")
        s = self._begin(sample, timestep, encoder_hidden_states, audio_features, speed_embeddings,
                        down_block_additional_residuals, mid_block_additional_residual, _ctx_kv=_ctx_kv,
                        halves_identical=_halves_identical, class_labels=class_labels)
        if self._reference_control is not None:
            self._reference_control._prepare(s.c, self)
        self._run_down(s)
        out = self._run_rest(s, return_rows=_return_rows)
        if _return_rows:
            return out
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    __call__ = forward

    # bank plumbing used by ReferenceAttentionControl / the pipeline
    def bank_order(self, fusion_blocks="midup"):
        return reference_block_order(self.spec, fusion_blocks)
