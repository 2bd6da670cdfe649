"""VideoNet - the reference's alternative denoising network (models/videonet.py:132-267) on MI355X.

The reference deep-copies a Stable-Diffusion 2-D UNet and replaces every attention slot (down blocks, mid block, up blocks - in
that order) by a `ReferenceConditionedAttentionBlock`:

    sam        SpatialAttentionModule  (:15-77)   the reference feature map concatenated along the WIDTH, q from the raw tokens
    cross_attn the slot's original Transformer2DModel (text cross-attention)
    tam        get_motion_module(C, 'Vanilla', {})  (:151-153)  the AnimateDiff temporal module with its ctor defaults: 8 heads,
               TWO transformer blocks, no positional encoding - over '(b t) c h w -> b c t h w'

Here VideoNet is an instance of the same launch sequence as UNet3DConditionModel with F = 1 (a 2-D UNet: per-frame GroupNorm in the
resnets, per-frame transformers - rows ((b t) h w, C)), whose transformer step is wrapped: emo_groupnorm / emo_gemm / emo_attention for
the sam (keys = [x | ref] tokens of a frame), the unchanged transformer, then the motion-module kernels on the same rows regrouped as
(b, t).  No new kernels; state-dict keys are the reference's (`unet.<block>.attentions.<j>.{cross_attn,sam,tam}....`,
tests/golden/videonet_wiring.json).  Same surface: `VideoNet(sd_unet, num_frames, batch_size)`, `update_reference_embeddings`,
`update_num_frames`, `update_skip_temporal_attn`, `forward(initial_noise, timesteps, reference_embeddings,
clip_condition_embeddings, skip_temporal_attn)`, `ref_cond_attn_blocks[i]` with `update_reference_tensor` / `forward`.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops
from ._lib import EmoHipError
from .spec import MotionSpec, param_shapes
from .unet import UNet3DConditionModel, _Ctx

SAM_HEADS = 8      # models/videonet.py:16 (num_heads default; the block passes embed_dim only, :149)
TAM_KWARGS = dict(heads=8, n_attn=2, pe_len=None, n_blocks=2)   # models/motionmodule.py:37-48 ctor defaults


class ReferenceConditionedAttentionBlock:
    """models/videonet.py:132-196 - a handle on one attention slot of a VideoNet (the kernels and the packed weights live in the net)."""

    def __init__(self, net: "VideoNet", spec, num_frames: int, skip_temporal_attn: bool = False):
        self._net, self.spec = net, spec
        self.num_frames, self.skip_temporal_attn = num_frames, skip_temporal_attn
        self.reference_tensor = None
        self._ref_rows = None

    def update_reference_tensor(self, reference_tensor: torch.Tensor):      # :158-159
        self.reference_tensor, self._ref_rows = reference_tensor, None

    def update_num_frames(self, num_frames: int):                            # :162-163
        self.num_frames = num_frames

    def _reference_rows(self, bt, H, W):
        net = self._net
        if self.reference_tensor is None:
            raise EmoHipError("ReferenceConditionedAttentionBlock: update_reference_tensor first (models/videonet.py:155-159)")
        if self._ref_rows is None:
            r = self.reference_tensor
            if tuple(r.shape) != (bt, self.spec.channels, H, W):
                raise ValueError(f"reference tensor {tuple(r.shape)} != hidden states {(bt, self.spec.channels, H, W)} (concatenated along the width, :42)")
            self._ref_rows = ops.ncfhw_to_rows(r.to(net.device).float().unsqueeze(2), net.dtype)
        return self._ref_rows

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, added_cond_kwargs=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None, return_dict=True):
        """(b*t, C, h, w) -> ((b*t, C, h, w),)   (:165-196; the extra arguments are the 2-D transformer's and unused, like upstream)."""
        net = self._net
        if net._w is None:
            raise EmoHipError("VideoNet: load_state_dict + .to('cuda') first (no CPU execution path)")
        bt, C, H, W = hidden_states.shape
        rows = ops.ncfhw_to_rows(hidden_states.to(net.device).float().unsqueeze(2), net.dtype)
        ctx = encoder_hidden_states.to(net.device)
        ctx_rows = ops.convert(ctx.float().reshape(-1, ctx.shape[2]), net.dtype)
        y = net._transformer(self.spec, rows, ctx_rows, ctx.shape[1], 1, _Ctx(bt, 1, H, W), H, W)
        return (ops.rows_to_ncfhw(y, bt, C, 1, H, W)[:, :, 0],)

    __call__ = forward


class VideoNet(UNet3DConditionModel):
    def __init__(self, sd_unet, num_frames: int = 24, batch_size: int = 2):
        """sd_unet: the 2-D Stable-Diffusion UNet to start from - a UNet3DConditionModel of this package WITHOUT motion modules (its
        F = 1 instance is the 2-D network), or its config dict.  Its weights, if loaded, are copied (the reference deep-copies, :205)."""
        cfg = dict(sd_unet.config) if hasattr(sd_unet, "config") else dict(sd_unet)
        if cfg.get("use_motion_module"):
            raise ValueError("VideoNet starts from a 2-D UNet (models/videonet.py:200): use_motion_module must be off - the temporal modules are the blocks' own `tam`")
        super().__init__(**cfg)
        self.batch_size, self.num_frames = batch_size, num_frames
        self.ref_cond_attn_blocks: List[ReferenceConditionedAttentionBlock] = []
        self._blocks = {}
        # models/videonet.py:216-234: down blocks, mid block, up blocks - the order reference_embeddings[i] is dealt in (:237-244)
        for blk in self.spec.down + [self.spec.mid] + self.spec.up:
            for a in blk.attentions:
                if a is None:
                    continue
                if a.channels % 32 or a.channels % SAM_HEADS or (a.channels // SAM_HEADS) % 8:
                    raise NotImplementedError(f"VideoNet: {a.channels} channels - sam / tam need GroupNorm(32) and 8 heads of a multiple of 8 channels")
                slot = a.prefix
                a.prefix, a.sam = slot + ".cross_attn", slot + ".sam"
                a.tam = MotionSpec(slot + ".tam", a.channels, TAM_KWARGS["heads"], TAM_KWARGS["n_attn"], TAM_KWARGS["pe_len"], TAM_KWARGS["n_blocks"])
                h = ReferenceConditionedAttentionBlock(self, a, num_frames)
                h.slot = slot
                self.ref_cond_attn_blocks.append(h)
                self._blocks[a.prefix] = h
        self._shapes = param_shapes(self.spec)
        src = getattr(sd_unet, "_master", None)
        if src:
            slots = [h.slot for h in self.ref_cond_attn_blocks]

            def moved(k):
                for s_ in slots:
                    if k.startswith(s_ + "."):
                        return s_ + ".cross_attn." + k[len(s_) + 1:]
                return k
            self.load_state_dict({moved(k): v for k, v in src.items()}, strict=False)

    # ---- state dict under the reference's `unet.` attribute name (VideoNet.unet, :205)
    def load_state_dict(self, sd, strict=True):
        return super().load_state_dict({(k[5:] if k.startswith("unet.") else k): v for k, v in sd.items()}, strict=strict)

    def state_dict(self):
        return {"unet." + k: v for k, v in super().state_dict().items()}

    # ---- models/videonet.py:236-250
    def update_reference_embeddings(self, reference_embeddings):
        if len(reference_embeddings) != len(self.ref_cond_attn_blocks):
            print("[!] WARNING - amount of input reference embeddings does not match number of modules in VideoNet")
        for i in range(len(self.ref_cond_attn_blocks)):
            self.ref_cond_attn_blocks[i].update_reference_tensor(reference_embeddings[i])

    def update_num_frames(self, num_frames):
        self.num_frames = num_frames
        for b in self.ref_cond_attn_blocks:
            b.update_num_frames(num_frames)

    def update_skip_temporal_attn(self, skip_temporal_attn):
        for b in self.ref_cond_attn_blocks:
            b.skip_temporal_attn = skip_temporal_attn

    # ---- the wrapped transformer step (ReferenceConditionedAttentionBlock.forward, :165-196) on NHWC rows
    def _sam(self, a, xr, rr, bt, H, W):
        """SpatialAttentionModule.forward (:39-77): xr, rr rows ((bt) h w, C) -> rows."""
        w, p, C = self._w, a.sam, a.channels
        # concat along the width (:42): token order (h, [x row | ref row]) per frame - an index / copy step
        cat = torch.cat([xr.view(bt, H, W, C), rr.view(bt, H, W, C)], dim=2).reshape(bt * H * 2 * W, C)
        pj = ops.gemm(ops.group_norm(cat, w[p + ".norm_in.g"], w[p + ".norm_in.b"], bt, 32, 1e-6, False), w[p + ".proj_in.w"], w[p + ".proj_in.b"])
        Lk = H * 2 * W
        q = ops.gemm(xr, w[p + ".to_q.w"], w[p + ".to_q.b"])                                   # q from the RAW tokens (:51,54)
        k = ops.gemm(pj, w[p + ".k.w"], w[p + ".k.b"])
        vt = ops.gemm(pj, w[p + ".v.w"], w[p + ".v.b"], transpose_rows=Lk, transpose_ld=(Lk + 7) // 8 * 8)
        d = C // SAM_HEADS
        att = ops.attention(q, k, vt, Lk, B=bt, Lq=H * W, heads=SAM_HEADS, d=d, scale=d ** -0.5)
        n1 = ops.layer_norm(ops.add(att, xr), w[p + ".norm1.g"], w[p + ".norm1.b"])            # :66
        n2 = ops.layer_norm(ops.add(n1, ops.gemm(n1, w[p + ".ffn.w"], w[p + ".ffn.b"])), w[p + ".norm2.g"], w[p + ".norm2.b"])
        return ops.gemm(n2, w[p + ".proj_out.w"], w[p + ".proj_out.b"], residual=xr)           # :74-77

    def _transformer(self, a, x, ctx_rows, ctx_len, ctx_div, c: _Ctx, H, W, out=None, ctx_kv=None, shared_half=False):
        blk = self._blocks[a.prefix]
        bt = c.B * c.F
        if not x.is_contiguous():
            x = x.contiguous()
        y = self._sam(a, x, blk._reference_rows(bt, H, W), bt, H, W)
        tam = not blk.skip_temporal_attn
        y = super()._transformer(a, y, ctx_rows, ctx_len, ctx_div, c, H, W, out=None if tam else out, ctx_kv=ctx_kv)
        if not tam:                                                                            # :188-189
            return y
        T = blk.num_frames
        if bt % T:
            raise ValueError(f"VideoNet: {bt} frames are not a multiple of num_frames={T} ('(b t) c h w -> b c t h w', :192)")
        return self._motion(a.tam, y, _Ctx(bt // T, T, H, W), H, W, out=out)                    # rows ((b t) h w, C) ARE the (b, t) grouping

    def forward(self, intial_noise, timesteps, reference_embeddings, clip_condition_embeddings, skip_temporal_attn=False):
        """models/videonet.py:252-267: (b*t, 4, h, w) noise, per-sample or scalar timesteps, one reference feature map (b*t, C_i, h_i, w_i)
        per block, text embeddings (b*t, L, D) -> (b*t, 4, h, w)."""
        self.update_reference_embeddings(reference_embeddings)
        self.update_skip_temporal_attn(skip_temporal_attn)
        if intial_noise.dim() != 4:
            raise ValueError("VideoNet takes the 2-D UNet's (b*t, c, h, w) input")
        return super().forward(intial_noise.unsqueeze(2), timesteps, clip_condition_embeddings, return_dict=False)[0][:, :, 0]

    __call__ = forward
