"""ControlNetModel - the pose / DensePose branch of the vendored pipeline
(magicanimate/models/controlnet.py:94-572).

The reference is a diffusers 2-D UNet encoder (down blocks + mid block; block arithmetic in third-party diffusers, parity
unpinned like the AppearanceEncoder) plus three pieces that ARE in the tree and are restated here:
  * ControlNetConditioningEmbedding (:49-91): conv_in 3x3, then (conv 3x3, conv 3x3 stride 2) pairs, SiLU after each,
    and a zero-initialised conv_out 3x3 - 512x512 pose image -> 64x64 x block_out_channels[0] features added to conv_in(sample);
  * the zero-convolutions (:209-243): a 1x1 conv on each of the 12 skip tensors and on the mid-block output;
  * forward (:450-567): residuals scaled by `conditioning_scale`.
Here it is the F=1, no-motion, no-up-path instance of the same kernels as the Backbone (same state-dict key names as the
reference module), the conditioning embedding runs on the implicit-GEMM conv kernel and the zero-convs are GEMMs with
`out_scale = conditioning_scale`.  Output format = the reference's: a list of (N, C, h, w) residuals and the mid residual,
ready for `UNet3DConditionModel.forward(down_block_additional_residuals=..., mid_block_additional_residual=...)` after the
pipeline's per-frame selection (EMOAnimationPipeline.py:514-540)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch

from . import ops
from .unet import UNet3DConditionModel, _round_up


@dataclass
class ControlNetOutput:
    down_block_res_samples: List[torch.Tensor]
    mid_block_res_sample: torch.Tensor


class ControlNetModel(UNet3DConditionModel):
    def __init__(self, **kwargs):
        kw = dict(kwargs)
        cc = tuple(kw.pop("conditioning_embedding_out_channels", (16, 32, 96, 256)) or (16, 32, 96, 256))
        order = kw.pop("controlnet_conditioning_channel_order", "rgb")
        for k in ("projection_class_embeddings_input_dim",):   # ctor kwargs of the reference with no effect on this path
            kw.pop(k, None)
        dbt = kw.get("down_block_types", ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"))
        if len(kw.get("block_out_channels", (320, 640, 1280, 1280))) != len(dbt):   # controlnet.py:127-130
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        kw["down_block_types"] = tuple(t.replace("2D", "3D") for t in dbt)   # the 2-D blocks are the F=1 case of the 3-D ones
        n = len(dbt)
        kw.setdefault("up_block_types", ("UpBlock3D",) + ("CrossAttnUpBlock3D",) * (n - 1))   # never built (no up path)
        kw.setdefault("unet_use_cross_frame_attention", False)
        kw.setdefault("unet_use_temporal_attention", False)
        kw["_has_out"] = False
        kw["_controlnet"] = cc
        super().__init__(**kw)
        self.controlnet_conditioning_channel_order = order
        self.conditioning_embedding_out_channels = cc

    @classmethod
    def from_unet(cls, unet, controlnet_conditioning_channel_order="rgb", conditioning_embedding_out_channels=(16, 32, 96, 256),
                  load_weights_from_unet=True):
        """controlnet.py:263-321: same configuration as `unet`; conv_in / time embedding / down blocks / mid block copied."""
        keys = ("in_channels", "flip_sin_to_cos", "freq_shift", "down_block_types", "only_cross_attention", "block_out_channels",
                "layers_per_block", "downsample_padding", "mid_block_scale_factor", "act_fn", "norm_num_groups", "norm_eps",
                "cross_attention_dim", "attention_head_dim", "use_linear_projection", "class_embed_type", "num_class_embeds",
                "upcast_attention", "resnet_time_scale_shift")
        net = cls(**{k: unet.config[k] for k in keys}, controlnet_conditioning_channel_order=controlnet_conditioning_channel_order,
                  conditioning_embedding_out_channels=conditioning_embedding_out_channels)
        if load_weights_from_unet:
            sd = {k: v for k, v in unet.state_dict().items()
                  if k.startswith(("conv_in.", "time_embedding.", "down_blocks.", "mid_block.")) and "motion_modules" not in k}
            net.load_state_dict(sd, strict=False)
        return net

    def _cond_embedding(self, cond, N, Hc, Wc):
        """ControlNetConditioningEmbedding.forward (controlnet.py:78-91) on NHWC rows."""
        w, dtp = self._w, self.dtype
        ce = "controlnet_cond_embedding"
        cin = cond.shape[1]
        x = ops.ncfhw_to_rows(cond.unsqueeze(2), dtp, cpad=_round_up(cin, 8))
        x, h, w_ = ops.conv3x3(x, w[ce + ".conv_in.w"], w[ce + ".conv_in.b"], N, Hc, Wc)
        x = ops.silu(x)
        for i in range(2 * (len(self.conditioning_embedding_out_channels) - 1)):
            x, h, w_ = ops.conv3x3(x, w[f"{ce}.blocks.{i}.w"], w[f"{ce}.blocks.{i}.b"], N, h, w_, stride=2 if i % 2 else 1)
            x = ops.silu(x)
        x, h, w_ = ops.conv3x3(x, w[ce + ".conv_out.w"], w[ce + ".conv_out.b"], N, h, w_)
        return x, h, w_

    def cond_embedding(self, controlnet_cond):
        """controlnet_cond_embedding(controlnet_cond) (controlnet.py:523) as NHWC rows + its (h, w): a function of the conditioning
        image only - not of the timestep or the latents - so a sampling loop computes it ONCE per clip and hands it to every step
        (`forward(..., _cond_rows=)`); the reference recomputes it inside every forward."""
        order = self.controlnet_conditioning_channel_order
        if order == "bgr":
            controlnet_cond = torch.flip(controlnet_cond, dims=[1])
        elif order != "rgb":
            raise ValueError(f"unknown `controlnet_conditioning_channel_order`: {order}")
        if controlnet_cond.dim() != 4:
            raise ValueError("controlnet_cond must be (N,3,H,W)")
        return self._cond_embedding(controlnet_cond.to(self.device).float(), controlnet_cond.shape[0], controlnet_cond.shape[2], controlnet_cond.shape[3])

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, return_dict=True, _cond_rows=None):
        """controlnet.py:450-567.  sample (N,4,h,w); controlnet_cond (N,3,8h,8w); returns the 12 down residuals (N,C,h',w')
        and the mid residual, each multiplied by `conditioning_scale`.  `_cond_rows` = a precomputed `cond_embedding()` of
        controlnet_cond (which may then be None)."""
        if attention_mask is not None or class_labels is not None or timestep_cond is not None:
            raise NotImplementedError("attention_mask / class_labels / timestep_cond are outside the hot path (always None in the pipeline)")
        if sample.dim() != 4:
            raise ValueError("ControlNetModel works on 2-D batches: sample (N,C,h,w), controlnet_cond (N,3,H,W)")
        N, _, h, w_ = sample.shape
        cond_rows, ch, cw = _cond_rows if _cond_rows is not None else self.cond_embedding(controlnet_cond)
        if (ch, cw) != (h, w_):
            raise ValueError(f"conditioning image maps to {ch}x{cw} but the latent is {h}x{w_}")
        s = self._begin(sample.unsqueeze(2), timestep, encoder_hidden_states, add_after_conv_in=cond_rows)
        self._run_down(s)
        x_mid = self._run_mid(s, s.x)
        w = self._w
        scale = float(conditioning_scale)
        down, hh, ww = [], h, w_
        res_hw = []
        # spatial size of each skip in push order: conv_in, the resnet sub-blocks of a level, then its downsampler output
        hw = [(h, w_)]
        ch_, cw_ = h, w_
        for blk in self.spec.down:
            hw += [(ch_, cw_)] * len(blk.resnets)
            if blk.sampler:
                ch_, cw_ = (ch_ + 2 - 3) // 2 + 1, (cw_ + 2 - 3) // 2 + 1
                hw.append((ch_, cw_))
        for k, (sk, (sh, sw)) in enumerate(zip(s.skips, hw)):
            r = ops.gemm(sk, w[f"controlnet_down_blocks.{k}.w"], w[f"controlnet_down_blocks.{k}.b"], out_scale=scale)
            down.append(ops.rows_to_ncfhw(r, N, r.shape[1], 1, sh, sw).squeeze(2))
        rm = ops.gemm(x_mid, w["controlnet_mid_block.w"], w["controlnet_mid_block.b"], out_scale=scale)
        mid = ops.rows_to_ncfhw(rm, N, rm.shape[1], 1, s.h, s.w).squeeze(2)
        if not return_dict:
            return (down, mid)
        return ControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid)

    __call__ = forward
