"""Oracle (test infrastructure): the denoising loop of EMOAnimationPipeline.__call__
(EMOAnimationPipeline.py:698-823), restated over the oracle UNet.  fp32 CPU.

Loop arithmetic pinned by tests/golden/loop_tiny.safetensors (a generator-side re-enactment of
the reference loop body around the reference's own UNet); the scheduler is parity-unpinned
(see scheduler_ref.py).
"""
from __future__ import annotations

import math

import torch

from .scheduler_ref import SchedulerRef, counter_normal, uniform_windows
from .unet_ref import round_banks_fp16, unet_forward


def denoise_loop(unet_sd, unet_cfg, ref_sd, ref_cfg, latents, ref_latents, text_embeddings, *,
                 scheduler: SchedulerRef, num_inference_steps=50, guidance_scale=7.5,
                 context_frames=16, context_stride=1, context_overlap=4, context_batch_size=1,
                 fusion_blocks="midup", seed=0, audio_features=None, speed_embeddings=None,
                 rank=0, world_size=1, return_eps=False, controlnet=None, motion_latents=None):
    """latents (1,4,F_tot,h,w); ref_latents (1,4,h,w); text_embeddings (2,L,D) = [uncond, cond].

    Per step (EMOAnimationPipeline.py:698-823):
      ReferenceNet write pass on ref_latents.repeat(2*cbs) at the current t      (:711-716)
      windows = uniform(0, steps, F_tot, context_frames, stride, overlap)       (:748-755)
      per window batch: x = cat(latents[:,:,c]).repeat(2); reader.update (fp16 banks); UNet;
                        noise_pred[:,:,c] += pred; counter[:,:,c] += 1            (:759-794)
      eps = uc + s*(c - uc) on noise_pred/counter                                (:812-814)
      latents = scheduler.step(eps, t, latents)                                  (:817)
    rank/world_size shard windows exactly as `global_context[rank::world_size]` (:757); with
    world_size>1 this function processes ALL ranks' windows in rank order (it is an oracle).
    controlnet = dict(sd=, cfg=, cond=(F_tot,3,H,W), scale=): per step the ControlNet runs per frame on the (scaled) latents
    with the cond text embedding, residuals are cached per frame (:718-746), then selected per window, reshaped
    '(b f) c h w -> b c f h w' and repeated for CFG (:514-540) and added inside the UNet (unet_controlnet.py:430-447)."""
    timesteps = scheduler.set_timesteps(num_inference_steps)
    cbs = context_batch_size
    # do_classifier_free_guidance = guidance_scale > 1.0 (:622).  Without it the UNet batch is the window batch (:759-763), the
    # text is the cond embedding alone and eps = noise_pred / counter.  (The reference's own lines do not run in that mode -
    # `pred_uc, pred_c = pred.chunk(2)` (:790) unpacks a one-row batch - and its reader is built with
    # do_classifier_free_guidance=True (:634), which would mask the first half of the FRAMES off the bank; the restatement follows
    # the evident intent, like the generator of tests/golden/loop_tiny.safetensors `ddim_nocfg`: no chunk, every row reads the bank.)
    cfg = guidance_scale > 1.0
    nbr = 2 if cfg else 1
    if not cfg:
        if motion_latents is not None or controlnet is not None or audio_features is not None:
            raise NotImplementedError("oracle: the no-CFG loop is restated for the plain path only")
        text_embeddings = text_embeddings[-1:]
    # (:631) torch.cat([text] * cbs) = [uc, c, uc, c, ...] - against latent rows [w0, w1, .., w0, w1, ..] and the reader's
    # uc mask [1, .., 0, ..] this pairs window j's rows with text row (branch * n + j) % 2 at cbs > 1: reference behaviour, kept
    text = torch.cat([text_embeddings] * cbs) if cbs > 1 else text_embeddings
    f_tot = latents.shape[2]
    eps_trace = []
    for si, t in enumerate(timesteps):
        noise_pred = torch.zeros(nbr, *latents.shape[1:])
        counter = torch.zeros(1, 1, f_tot, 1, 1)
        if motion_latents is None:
            ref_in = ref_latents.repeat(nbr * cbs, 1, 1, 1).unsqueeze(2)  # F=1 instance
            _, written = unet_forward(ref_sd, ref_cfg, ref_in, t, text, bank_mode="write", fusion_blocks=fusion_blocks)
        else:
            # motion-frame conditioning (design, no reference behaviour): [reference image, motion frames] through
            # the ReferenceNet with the cond text; their LN1 rows concatenated along tokens form ONE bank row, shared by every copy
            imgs = torch.cat([ref_latents.reshape(1, *ref_latents.shape[-3:]), motion_latents]).unsqueeze(2)
            _, w1 = unet_forward(ref_sd, ref_cfg, imgs, t, text_embeddings[1:].expand(imgs.shape[0], -1, -1), bank_mode="write",
                                 fusion_blocks=fusion_blocks)
            written = {p_: v.reshape(1, -1, v.shape[-1]).repeat(2 * cbs, 1, 1) for p_, v in w1.items()}
        banks = round_banks_fp16(written)
        windows = uniform_windows(0, num_inference_steps, f_tot, context_frames, context_stride, context_overlap)
        nb = math.ceil(len(windows) / cbs)
        batches = [windows[i * cbs:(i + 1) * cbs] for i in range(nb)]
        cn_cache = None
        if controlnet is not None:
            from .controlnet_ref import controlnet_forward
            cn_cache = {}
            for c in uniform_windows(0, num_inference_steps, f_tot, context_frames, context_stride, 0):   # overlap 0 (:723-725)
                xin = scheduler.scale_model_input(latents[:, :, c], t)[0].permute(1, 0, 2, 3)              # (f, c, h, w)
                dn, md = controlnet_forward(controlnet["sd"], controlnet["cfg"], xin, t, text_embeddings[1:].repeat(len(c), 1, 1),
                                            controlnet["cond"][c], controlnet.get("scale", 1.0))
                for j, k in enumerate(c):
                    cn_cache[k] = ([d[j:j + 1] for d in dn], md[j:j + 1])
        for r in range(world_size):
            for context in batches[r::world_size]:
                x = torch.cat([latents[:, :, c] for c in context]).repeat(nbr, 1, 1, 1, 1)
                x = scheduler.scale_model_input(x, t)
                b, _, f, _, _ = x.shape
                uc_rows = torch.zeros(b * f, dtype=torch.bool)
                if cfg:
                    uc_rows[: (b // 2) * f] = True
                af = None
                if audio_features is not None:  # (F_tot, L_a, D) per-frame ctx; uc rows get zeros
                    cond = torch.cat([audio_features[c] for c in context])
                    af = torch.cat([torch.zeros_like(cond), cond])
                ckw = {}
                if cn_cache is not None:   # select_controlnet_res_samples (:514-540)
                    frames = [k for c in context for k in c]
                    n_res = len(cn_cache[frames[0]][0])

                    def to5(ts):
                        y = torch.cat(ts)                                                     # ((b f), C, h, w), b = len(context)
                        y = y.reshape(len(context), f, *y.shape[1:]).permute(0, 2, 1, 3, 4)    # b c f h w
                        return y.repeat(2, 1, 1, 1, 1)
                    ckw = dict(down_block_additional_residuals=[to5([cn_cache[k][0][i] for k in frames]) for i in range(n_res)],
                               mid_block_additional_residual=to5([cn_cache[k][1] for k in frames]))
                pred = unet_forward(unet_sd, unet_cfg, x, t, text[:b], bank_mode="read", banks=banks,
                                    uc_rows=uc_rows, fusion_blocks=fusion_blocks, audio_features=af,
                                    speed_embeddings=speed_embeddings, **ckw)
                if cfg:
                    pred_uc, pred_c = pred.chunk(2)
                    pred = torch.stack([pred_uc, pred_c])
                else:
                    pred = pred.unsqueeze(0)
                for j, c in enumerate(context):
                    # reference (:792-794): noise_pred[:, :, c] = noise_pred[:, :, c] + pred[:, j]; counter likewise.  A wrapped
                    # window at context_stride > 1 lists a frame twice; torch leaves WHICH of the two writes of an index
                    # assignment survives undefined (serial CPU: the last one; parallel CPU / CUDA index_put: a race), so
                    # the restatement pins the serial outcome explicitly - one occurrence per frame, the last
                    last = {k: i for i, k in enumerate(c)}
                    keep = [i for i, k in enumerate(c) if last[k] == i]
                    frames = [c[i] for i in keep]
                    noise_pred[:, :, frames] = noise_pred[:, :, frames] + pred[:, j][:, :, keep]
                    counter[:, :, frames] = counter[:, :, frames] + 1
        if cfg:
            uc, cc = (noise_pred / counter).chunk(2)
            eps = uc + guidance_scale * (cc - uc)
        else:
            eps = noise_pred / counter
        if return_eps:
            eps_trace.append(eps.clone())
        z = None
        if scheduler.kind == "ddpm" or scheduler.eta > 0:
            z = counter_normal(seed, si, latents.numel()).reshape(latents.shape)
        latents = scheduler.step(eps, t, latents, noise=z)
    return (latents, eps_trace) if return_eps else latents
