"""AutoencoderKL - the VAE either side of the sampling loop on MI355X.

The reference takes the VAE from third-party `diffusers` (`AutoencoderKL`, not in tree; call sites
EMOAnimationPipeline.py:291-307 `decode_latents` - latents / 0.18215, one frame per `vae.decode` call,
`(video / 2 + 0.5).clamp(0, 1)` - and :402-414 `vae.encode(image).latent_dist.mean * 0.18215` for the reference image).
PARITY UNPINNED at this boundary like every diffusers block: the arithmetic is restated from the published SD-1.x
AutoencoderKL structure (oracle/vae_ref.py) under diffusers' state-dict key names, so a real `vae/diffusion_pytorch_model.bin`
loads; both the current attention key names (`to_q/to_k/to_v/to_out.0`) and the older ones (`query/key/value/proj_attn`) are
accepted.

    decoder: post_quant_conv 1x1 -> conv_in 3x3 (4 -> 512) -> mid [resnet, 1-head attention over h*w tokens, resnet]
             -> 4 up blocks of 3 resnets (512, 512, 256, 128), nearest x2 + conv3x3 after the first three
             -> GroupNorm(32, eps 1e-6) + SiLU -> conv_out 3x3 (128 -> 3)
    encoder: conv_in 3x3 (3 -> 128) -> 4 down blocks of 2 resnets (128, 256, 512, 512), pad (0,1,0,1) + conv3x3 stride 2 after
             the first three -> mid -> GroupNorm + SiLU -> conv_out 3x3 (512 -> 8) -> quant_conv 1x1 -> (mean, logvar)

Same kernels as the UNet at F = 1: NHWC rows, implicit-GEMM / halo 3x3 convs, per-image GroupNorm(+SiLU), MFMA GEMMs.  The
mid-block attention has ONE head of 512 channels - outside the flash kernel's head-dim classes - and runs as Q.K^T GEMM ->
emo_softmax_rows -> P.V GEMM per image (a 4096 x 4096 bf16 score matrix is 32 MB).  No torch arithmetic, no CPU fallback.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import ops
from ._lib import EmoHipError

VAE_DEFAULTS = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                    norm_num_groups=32, scaling_factor=0.18215, sample_size=512)

_OLD_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """latent_dist of AutoencoderKL.encode: parameters (B, 2*latent, h, w) = [mean | logvar]; logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor):
        self.mean, logvar = parameters.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution

    def __getitem__(self, k):   # diffusers BaseOutput: the reference reads `vae.encode(x)['latent_dist']` (EMOAnimationPipeline.py:411)
        return self.latent_dist if k in ("latent_dist", 0) else getattr(self, k)


def _round_up(x, m):
    return (x + m - 1) // m * m


def vae_param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """state-dict keys / shapes of diffusers' AutoencoderKL for `cfg` (module registration order)."""
    boc, L, lat = tuple(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["latent_channels"]
    d = OrderedDict()

    def conv(p, co, ci, k=3):
        d[p + ".weight"], d[p + ".bias"] = (co, ci, k, k), (co,)

    def resnet(p, ci, co):
        d[p + ".norm1.weight"], d[p + ".norm1.bias"] = (ci,), (ci,)
        conv(p + ".conv1", co, ci)
        d[p + ".norm2.weight"], d[p + ".norm2.bias"] = (co,), (co,)
        conv(p + ".conv2", co, co)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def mid(p, c):
        attn = p + ".attentions.0"
        d[attn + ".group_norm.weight"], d[attn + ".group_norm.bias"] = (c,), (c,)
        for n_ in ("to_q", "to_k", "to_v", "to_out.0"):
            d[f"{attn}.{n_}.weight"], d[f"{attn}.{n_}.bias"] = (c, c), (c,)
        resnet(p + ".resnets.0", c, c)
        resnet(p + ".resnets.1", c, c)

    conv("encoder.conv_in", boc[0], cfg["in_channels"])
    ci = boc[0]
    for i, co in enumerate(boc):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
        ci = co
    mid("encoder.mid_block", boc[-1])
    d["encoder.conv_norm_out.weight"], d["encoder.conv_norm_out.bias"] = (boc[-1],), (boc[-1],)
    conv("encoder.conv_out", 2 * lat, boc[-1])
    conv("decoder.conv_in", boc[-1], lat)
    rb = tuple(reversed(boc))
    ci = rb[0]
    for i, co in enumerate(rb):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci if j == 0 else co, co)
        if i != len(rb) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
        ci = co
    mid("decoder.mid_block", rb[0])
    d["decoder.conv_norm_out.weight"], d["decoder.conv_norm_out.bias"] = (rb[-1],), (rb[-1],)
    conv("decoder.conv_out", cfg["out_channels"], rb[-1])
    conv("quant_conv", 2 * lat, 2 * lat, 1)
    conv("post_quant_conv", lat, lat, 1)
    return d


class AutoencoderKL:
    def __init__(self, **kwargs):
        cfg = dict(VAE_DEFAULTS)
        for k, v in kwargs.items():
            if k in cfg:
                cfg[k] = v
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = SimpleNamespace(**cfg)
        self._cfg = cfg
        self._shapes = vae_param_shapes(cfg)
        self.dtype, self.device = torch.float32, torch.device("cpu")
        self._master, self._w = None, None

    # ------------------------------------------------------------------ torch-module-like surface
    def eval(self):
        return self

    def requires_grad_(self, *_):
        return self

    def state_dict(self):
        if self._master is None:
            raise EmoHipError("no weights loaded")
        return {k: v.detach().cpu() for k, v in self._master.items()}

    def load_state_dict(self, sd, strict=True):
        sd = dict(sd)
        for k in list(sd):   # older diffusers attention key names; Linear weights stored as 1x1 convs in the oldest checkpoints
            for old, new in _OLD_ATTN.items():
                tag = f".attentions.0.{old}."
                if tag in k:
                    sd[k.replace(tag, f".attentions.0.{new}.")] = sd.pop(k)
        for k, v in list(sd.items()):
            if ".attentions.0." in k and k.endswith(".weight") and v.dim() == 4:
                sd[k] = v.reshape(v.shape[0], v.shape[1])
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
        self._master, self._w = master, None
        if all(k in master for k in self._shapes):
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
            if dtype not in (torch.float32, torch.bfloat16, torch.float16):
                raise EmoHipError(f"compute dtype {dtype} not supported (float32 | bfloat16 | float16)")
            self.dtype = dtype
        if device is not None:
            self.device = torch.device(device)
        if self._master is not None and all(k in self._master for k in self._shapes):
            self._pack()
        return self

    def _pack(self):
        if self.device.type != "cuda":
            return
        dev, dtp = self.device, self.dtype
        m = {k: v.to(dev) for k, v in self._master.items()}
        self._master = m
        w = {}
        for k, t in m.items():
            if k.endswith(".bias") or t.dim() == 1:
                w[k] = t.float().contiguous()
            elif t.dim() == 4 and t.shape[-1] == 3:            # OIHW -> [Cout][ky][kx][Cin_pad]
                co, ci = t.shape[0], t.shape[1]
                cip = _round_up(ci, 8)
                o = torch.zeros(co, 3, 3, cip, device=dev, dtype=torch.float32)
                o[..., :ci] = t.permute(0, 2, 3, 1)
                w[k] = o.reshape(co, 9 * cip).to(dtp).contiguous()
            else:                                              # Linear / 1x1 conv -> [N][K]
                w[k] = t.reshape(t.shape[0], -1).to(dtp).contiguous()
        # 1x1 convs on 4- / 8-channel latents: pad K to the 16-byte vector
        for k in ("quant_conv.weight", "post_quant_conv.weight"):
            t = w[k]
            kp = _round_up(t.shape[1], ops.vec(dtp))
            if kp != t.shape[1]:
                o = torch.zeros(t.shape[0], kp, device=dev, dtype=dtp)
                o[:, :t.shape[1]] = t
                w[k] = o.contiguous()
        self._w = w

    # ------------------------------------------------------------------ blocks
    def _need(self):
        if self._w is None:
            raise EmoHipError("AutoencoderKL: weights not loaded / model not on a HIP device (no CPU execution path)")

    def _gn(self, p, x, n_img, silu):
        return ops.group_norm(x, self._w[p + ".weight"], self._w[p + ".bias"], n_img, self._cfg["norm_num_groups"], 1e-6, silu)

    def _conv(self, p, x, n_img, H, W, **kw):
        return ops.conv3x3(x, self._w[p + ".weight"], self._w[p + ".bias"], n_img, H, W, **kw)

    def _resnet(self, p, x, n_img, H, W):
        """diffusers ResnetBlock2D without a time embedding: GN -> SiLU -> conv -> GN -> SiLU -> conv, + (1x1 conv) shortcut"""
        w = self._w
        h = self._gn(p + ".norm1", x, n_img, True)
        h, _, _ = self._conv(p + ".conv1", h, n_img, H, W)
        h = self._gn(p + ".norm2", h, n_img, True)
        sc = ops.gemm(x, w[p + ".conv_shortcut.weight"], w[p + ".conv_shortcut.bias"]) if (p + ".conv_shortcut.weight") in w else x
        out, _, _ = self._conv(p + ".conv2", h, n_img, H, W, residual=sc)
        return out

    def _mid_attention(self, p, x, n_img, HW):
        """one head over the h*w tokens of each image: GN -> q, k, V^T projections -> softmax(q k^T / sqrt(C)) v -> out + x"""
        w = self._w
        C_ = x.shape[1]
        h = self._gn(p + ".group_norm", x, n_img, False)
        # q leaves its projection already SCALED by 1/sqrt(C): unscaled Q.K^T over 512 channels overflows fp16 (the known SD-VAE
        # inf -> NaN) and costs bf16 ~22x the absolute precision of the logits; scaling q keeps every stored tensor in range
        scale = float(C_) ** -0.5
        q = ops.gemm(h, w[p + ".to_q.weight"], w[p + ".to_q.bias"], out_scale=scale)
        k = ops.gemm(h, w[p + ".to_k.weight"], w[p + ".to_k.bias"])
        ld = _round_up(HW, 8)
        vt = ops.gemm(h, w[p + ".to_v.weight"], w[p + ".to_v.bias"], transpose_rows=HW, transpose_ld=ld)   # (n, C, ld)
        if ld != HW:
            vt[:, :, HW:].zero_()
        att = torch.empty(n_img * HW, C_, device=x.device, dtype=x.dtype)
        for i in range(n_img):
            s = ops.gemm(q[i * HW:(i + 1) * HW], k[i * HW:(i + 1) * HW].contiguous())                  # (HW, HW) scaled scores
            pm = torch.zeros(HW, ld, device=x.device, dtype=x.dtype) if ld != HW else torch.empty(HW, HW, device=x.device, dtype=x.dtype)
            ops.softmax_rows(s, 1.0, out=pm[:, :HW])
            ops.gemm(pm, vt[i], out=att[i * HW:(i + 1) * HW])
        return ops.gemm(att, w[p + ".to_out.0.weight"], w[p + ".to_out.0.bias"], residual=x)

    def _mid(self, p, x, n_img, H, W):
        x = self._resnet(p + ".resnets.0", x, n_img, H, W)
        x = self._mid_attention(p + ".attentions.0", x, n_img, H * W)
        return self._resnet(p + ".resnets.1", x, n_img, H, W)

    # ------------------------------------------------------------------ decode / encode
    def _decode_rows(self, z):
        """z (n, latent, h, w) f32 -> decoded NHWC rows (n*8h*8w, 3) in the compute dtype"""
        self._need()
        w, cfg, dtp = self._w, self._cfg, self.dtype
        n, lat, H, W = z.shape
        # images ride the frame axis of the layout kernel: (1, lat, n, H, W) -> rows ((n) h w, lat padded to the 16-byte vector)
        x = ops.ncfhw_to_rows(z.to(self.device).float().permute(1, 0, 2, 3).unsqueeze(0), dtp, cpad=_round_up(lat, ops.vec(dtp)))
        xin = torch.zeros(x.shape[0], _round_up(lat, 8), device=x.device, dtype=dtp)     # conv_in reads 8-channel rows
        ops.gemm(x, w["post_quant_conv.weight"], w["post_quant_conv.bias"], out=xin[:, :lat])
        x, _, _ = self._conv("decoder.conv_in", xin, n, H, W)
        x = self._mid("decoder.mid_block", x, n, H, W)
        nb = len(cfg["block_out_channels"])
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, n, H, W)
            if i != nb - 1:
                x, H, W = self._conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x, n, H, W, upsample2x=True)
        x = self._gn("decoder.conv_norm_out", x, n, True)
        x, _, _ = self._conv("decoder.conv_out", x, n, H, W)
        return x, H, W

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """AutoencoderKL.decode: z (n, 4, h, w) -> sample (n, 3, 8h, 8w) f32"""
        rows, H, W = self._decode_rows(z)
        n = z.shape[0]
        y = ops.rows_to_ncfhw(rows, 1, self._cfg["out_channels"], n, H, W)[0].permute(1, 0, 2, 3).contiguous()
        return DecoderOutput(sample=y) if return_dict else (y,)

    @torch.no_grad()
    def decode_video(self, latents, frames_per_call=4):
        """decode_latents (EMOAnimationPipeline.py:291-307): latents (b, 4, f, h, w) -> video (b, 3, f, 8h, 8w) f32 in [0, 1].
        The reference decodes one frame per call; frames are batched here (`frames_per_call`) - the network is per-image."""
        b, c4, f, h, w = latents.shape
        lat = (latents.to(self.device).float() * (1.0 / self._cfg["scaling_factor"])).permute(0, 2, 1, 3, 4).reshape(b * f, c4, h, w)
        out = torch.empty(b, self._cfg["out_channels"], f, 8 * h, 8 * w, device=self.device, dtype=torch.float32)
        for i0 in range(0, b * f, frames_per_call):
            i1 = min(i0 + frames_per_call, b * f)
            rows, H, W = self._decode_rows(lat[i0:i1])
            vid = ops.rows_to_video(rows, 1, self._cfg["out_channels"], i1 - i0, H, W)      # (1, 3, n, H, W), (x/2 + .5).clamp(0, 1)
            for k in range(i0, i1):
                out[k // f, :, k % f] = vid[0, :, k - i0]
        return out

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """AutoencoderKL.encode: x (n, 3, H, W) in [-1, 1] -> latent_dist with .mean / .sample() (n, 4, H/8, W/8) f32"""
        self._need()
        w, cfg, dtp = self._w, self._cfg, self.dtype
        n, ci, H, W = x.shape
        rows = ops.ncfhw_to_rows(x.to(self.device).float().unsqueeze(0).permute(0, 2, 1, 3, 4).contiguous(), dtp, cpad=_round_up(ci, 8))
        h, _, _ = self._conv("encoder.conv_in", rows, n, H, W)
        nb = len(cfg["block_out_channels"])
        for i in range(nb):
            for j in range(cfg["layers_per_block"]):
                h = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", h, n, H, W)
            if i != nb - 1:   # F.pad(x, (0, 1, 0, 1)) + conv3x3 stride 2 padding 0
                h, H, W = self._conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", h, n, H, W, stride=2, pad=0)
        h = self._mid("encoder.mid_block", h, n, H, W)
        h = self._gn("encoder.conv_norm_out", h, n, True)
        h, _, _ = self._conv("encoder.conv_out", h, n, H, W)
        mom = ops.gemm(h, w["quant_conv.weight"], w["quant_conv.bias"])
        params = ops.rows_to_ncfhw(mom, 1, 2 * cfg["latent_channels"], n, H, W)[0].permute(1, 0, 2, 3).contiguous()
        dist = DiagonalGaussianDistribution(params)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)
