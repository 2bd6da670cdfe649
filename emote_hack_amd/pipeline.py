"""EMOAnimationPipeline - the sampling loop of the reference (EMOAnimationPipeline.py:543-840) on
MI355X.  Same `__call__` signature; the hot loop (`:698-823`) is re-designed:

  * per step: Backbone read passes over this rank's work units -> (all_gather of the eps slices) -> window average + CFG +
    scheduler step fused in ONE HIP kernel (emo_cfg_step) on f32 master latents;
  * the ReferenceNet banks depend on the timestep only (never on the latents), so the write pass is hoisted out of the
    step: T timesteps (reference_group, default 10) go through ONE batched pass (M is T times larger: the batch-1 pass of
    the reference is a small-M, launch-bound workload), their K / V^T projections with the Backbone's attn1 weights are
    computed once, kept resident in HBM (2 groups x 28 MB x T at 512^2) and selected per step by a device-side row index;
    the pass of group g+1 runs on a second HIP stream while the Backbone works through group g.  With world_size > 1 the
    ranks deal a group's timesteps among themselves and swap the banks with one RCCL all_gather (north_star: "all-gather
    over xGMI to broadcast ReferenceNet features");
  * work units are (context window x CFG branch): rank r owns U[r::world_size]; a UNet call batches the
    rank's next 2*context_batch_size units, uncond first - at world_size 1 exactly the reference's batch (:759-763).  The
    reference's gather-to-root + broadcast (:796-821) is replaced by ONE all_gather of the eps slices, after which every rank
    accumulates all units in the same order and runs the (deterministic, counter-based-noise) sampler step redundantly -
    no broadcast, bit-identical latents on all ranks;
  * the attn2 K / V^T projections of the text / audio context are computed once per clip, not per step.

Caller-supplied where out of scope: CLIP text encoder (`text_embeddings=`), VAE (`vae=` object with
encode / decode_video, see emote_hack_amd/vae.py), wav2vec (`audio_features=`, windowed by
emote_hack_amd.conditioning.audio_windows), `speed_embeddings=`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, List, Optional, Union

import torch

from . import ops
from .context import get_context_scheduler
from .reference_control import ReferenceAttentionControl
from .scheduler import DDIMScheduler, DDPMScheduler


@dataclass
class AnimationPipelineOutput:  # EMOAnimationPipeline.py:79-81
    videos: Union[torch.Tensor, "object"]


def _default(v, d):
    return d if v is None else v


# ---- magicanimate/utils/util.py:116-140 (the frame interpolation `interpolate_latents` uses)
tensor_interpolation = None


def get_tensor_interpolation_method():
    return tensor_interpolation


def set_tensor_interpolation_method(is_slerp):
    global tensor_interpolation
    tensor_interpolation = slerp if is_slerp else linear


def linear(v1, v2, t):
    return (1.0 - t) * v1 + t * v2


def slerp(v0: torch.Tensor, v1: torch.Tensor, t: float, DOT_THRESHOLD: float = 0.9995) -> torch.Tensor:
    """spherical interpolation of two tensors taken as single vectors; linear when they are nearly parallel (util.py:128-140)"""
    dot = ((v0 / v0.norm()) * (v1 / v1.norm())).sum()
    if dot.abs() > DOT_THRESHOLD:
        return (1.0 - t) * v0 + t * v1
    omega = dot.acos()
    return (((1.0 - t) * omega).sin() * v0 + (t * omega).sin() * v1) / omega.sin()


class EMOAnimationPipeline:
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, controlnet=None, scheduler=None):
        """EMOAnimationPipeline.py:87-130.  The ctor forces steps_offset=1 / clip_sample=False on the
        scheduler it is given, like the reference."""
        if unet is None or scheduler is None:
            raise ValueError("unet and scheduler are required")
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.controlnet, self.scheduler = unet, controlnet, scheduler
        # EMOAnimationPipeline.py:105-117: ANY scheduler whose config has the key, not DDIM only - a DDPMScheduler handed to the
        # pipeline runs [981, ..., 1] on 50 of 1000 steps, like a diffusers DDPMScheduler would under the reference ctor
        if getattr(scheduler.config, "steps_offset", 1) != 1:
            scheduler.config.steps_offset = 1
        if getattr(scheduler.config, "clip_sample", False):
            scheduler.config.clip_sample = False
        self.vae_scale_factor = 8
        self.device = unet.device
        self._plan_cache = None   # (plan key, prepared state) of the last `denoise(reuse_state=True)` / `__call__`

    @property
    def _execution_device(self):
        return self.unet.device

    def check_inputs(self, prompt, height, width, callback_steps):  # EMOAnimationPipeline.py:326-339
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def prepare_denoise(self, latents, ref_image_latents, text_embeddings, *, appearance_encoder, num_inference_steps=50,
                        guidance_scale=7.5, eta=0.0, context_frames=16, context_stride=1, context_overlap=4,
                        context_batch_size=1, context_schedule="uniform", audio_features=None, speed_embeddings=None, seed=0,
                        fusion_blocks="midup", dist=False, rank=0, world_size=1, return_eps=False, use_graphs=None,
                        controlnet=None, controlnet_cond=None, controlnet_conditioning_scale=1.0, reference_group=10,
                        reference_lookahead=None, motion_latents=None, text_pairing="reference", _emulate_rank=None):
        """Set up the loop state (EMOAnimationPipeline.py:628-696).  latents f32 (1,4,F_tot,h,w);
        ref_image_latents (1,4,h,w); text_embeddings (2,L,D) = [uncond, cond].
        reference_group = T: ReferenceNet timesteps computed per batched pass (1 = the reference's per-step order).
        use_graphs: None (default) = HIP-graph replay of the UNet / ReferenceNet passes whenever the model lives on a HIP device
        (the path bench.py measures; ~1700 launches per step leave the host), False = every kernel launched from Python.
        The state splits into a PLAN (geometry, buffers, captured graphs - everything above `_bind_inputs`) and the INPUTS
        (latents, reference image, text / audio / speed conditioning, ControlNet images, guidance scale, seed), which are copied
        into the plan's buffers in place: `reset_denoise` re-arms a prepared state for another clip of the same geometry.
        motion_latents (n_m, 4, h, w): latents of the previous clip's last frames - they go through the ReferenceNet next to the
        reference image and their LN1 features join the banks as extra tokens (see denoise_chained).
        text_pairing (matters at context_batch_size > 1 only): "reference" = the upstream row pairing, literally (see `unit_tv`
        below: odd windows of a batch run their uncond row under the cond text and vice versa, which cancels guidance for them);
        "branch" = the evidently intended one, every uncond row under the uncond text, every cond row under the cond text and the
        cond-text bank - the same function as context_batch_size 1, batched.
        _emulate_rank=(r, n): MEASUREMENT aid (bench.py --emulate-rank): the work of rank r of an n-rank job - its units, its share of every
        ReferenceNet group, the projection of the whole group - in ONE process without collectives (the other ranks' eps slices are
        missing, so the latents are not a sample of anything): a rank's critical path per step on an otherwise idle GPU."""
        if text_pairing not in ("reference", "branch"):
            raise ValueError(f"text_pairing must be 'reference' or 'branch', got {text_pairing!r}")
        unet, sch = self.unet, self.scheduler
        dev = unet.device
        # `do_classifier_free_guidance = guidance_scale > 1.0` (:622).  Without it the UNet batch is the window batch (:759-763
        # `.repeat(1)`), the text is the cond embedding alone, every row reads the bank and eps = noise_pred / counter.  (The
        # reference's own lines do not run in that mode: `pred_uc, pred_c = pred.chunk(2)` (:790) unpacks a one-row batch, and its
        # reader is built with do_classifier_free_guidance=True (:634), which would mask half of the FRAMES off the bank; this
        # follows the evident intent, pinned by the `ddim_nocfg` loop golden.)
        cfg = self._do_cfg(guidance_scale)
        if latents.shape[0] != 1:
            raise ValueError("batch_size must be 1 (EMOAnimationPipeline.py:641-642); run clips as separate calls")
        text_embeddings = self._text_rows(text_embeddings, cfg)
        st = SimpleNamespace()
        st.cfg = cfg
        st.cbs = cbs = int(context_batch_size)
        if cbs < 1:
            raise ValueError("context_batch_size must be >= 1")
        st.latents = torch.empty(tuple(latents.shape), device=dev, dtype=torch.float32)
        _, st.C4, st.f_tot, st.h, st.w = st.latents.shape
        st.HW = st.h * st.w
        # text rows indexed by VARIANT: 0 = uncond, 1 = cond (without guidance both name the cond row)
        st.text = torch.empty(2, *text_embeddings.shape[1:], device=dev, dtype=torch.float32)
        st.appearance_encoder = appearance_encoder
        st.writer = ReferenceAttentionControl(appearance_encoder, do_classifier_free_guidance=True, mode="write",
                                              batch_size=cbs, fusion_blocks=fusion_blocks)      # :633
        st.reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=cfg, mode="read", batch_size=cbs,
                                              fusion_blocks=fusion_blocks)                        # :634
        st.num_inference_steps = num_inference_steps
        st.scheduler = sch
        st.timesteps = list(sch.set_timesteps(num_inference_steps))
        n_steps = len(st.timesteps)
        # ReferenceNet images = [reference, motion frames] (Net.py:56-72 intent; junk/EMo-write-up.txt:104-110)
        st.n_ref_images = 1 + (0 if motion_latents is None else self._motion_rows(motion_latents).shape[0])
        st.ref_lat = torch.empty(st.n_ref_images, st.C4, st.h, st.w, device=dev, dtype=torch.float32)
        scheduler_fn = get_context_scheduler(context_schedule)
        # the reference recomputes the (step-independent: step arg is always 0) window list every step (:748-755)
        st.windows = [list(map(int, c)) for c in scheduler_fn(0, num_inference_steps, st.f_tot, context_frames, context_stride, context_overlap)]
        nf = len(st.windows[0])
        if any(len(c) != nf for c in st.windows):
            raise ValueError("context windows of unequal length")
        st.nf = nf
        nb = math.ceil(len(st.windows) / cbs)
        st.global_context = [st.windows[i * cbs:(i + 1) * cbs] for i in range(nb)]      # the reference's window batches (:752-755)
        # `noise_pred[:, :, c] = noise_pred[:, :, c] + pred` (:792-794) with a frame listed twice in c (wrapped windows at
        # context_stride > 1) keeps ONE occurrence (the last write wins): positions of earlier duplicates are marked -1
        st.frame_idx = []
        for c in st.windows:
            last = {k: j for j, k in enumerate(c)}
            st.frame_idx.append(torch.tensor([k if last[k] == j else -1 for j, k in enumerate(c)], dtype=torch.int32, device=dev))
        # ---- work units: U = [(window, branch)], branch 0 = uncond, 1 = cond; rank r owns U[r::world].  A UNet
        # call batches up to 2*cbs of the rank's units, uncond units first (they skip the reference banks, :243-256) - at
        # world_size 1 that is exactly the reference's [uc x cbs, c x cbs] batch (:759-763).
        st.dist = bool(dist)
        st.rank, st.world_size = (int(rank), int(world_size)) if st.dist else (0, 1)
        st.emulate = _emulate_rank is not None
        if st.emulate:
            if st.dist:
                raise ValueError("_emulate_rank replaces dist=True")
            st.rank, st.world_size = int(_emulate_rank[0]), int(_emulate_rank[1])
        # (branch-major order: with as many windows as ranks a rank owns BOTH branches of one window - balanced, the cond
        # branch costs ~8 % more - and with twice as many ranks as windows, BASELINE configs[3], every rank owns one unit)
        st.branches = (0, 1) if cfg else (1,)
        st.np_slot = {br: i for i, br in enumerate(st.branches)}           # branch -> plane of the noise_pred accumulator
        st.units = [(w, br) for br in st.branches for w in range(len(st.windows))]
        # Text / bank VARIANT of a unit.  The reference stacks `torch.cat([text_embeddings] * context_batch_size)` (:631) =
        # [uc, c, uc, c, ..] against latent rows [w0, w1, .., w0, w1, ..] (:759-763) and a uc mask [1, .., 0, ..]
        # (mutual_self_attention.py:186-197): row r = branch * n + j of a batch of n windows is paired with text row r % 2, and a
        # cond row reads bank row r - written by the ReferenceNet under text row r % 2 as well (:711-716).  At
        # context_batch_size 1 that is variant = branch; at context_batch_size > 1 it is NOT (window 0's cond row runs under the
        # uncond text and the uncond-text bank) - reference behaviour, reproduced (golden `ddim_cbs2`); text_pairing="branch"
        # pairs by branch instead.
        st.unit_tv = {}
        pos = 0
        for wb in st.global_context:
            n = len(wb)
            for j in range(n):
                for br in st.branches:
                    st.unit_tv[(pos + j, br)] = (br if text_pairing == "branch" else (br * n + j) % 2) if cfg else 1
            pos += n
        st.bank_variants = sorted({st.unit_tv[u] for u in st.units if u[1] == 1})   # the same on every rank
        mine = st.units[st.rank::st.world_size]
        st.n_slots = -(-len(st.units) // st.world_size)                    # units per rank, padded
        st.calls = []
        # which of the rank's units share a UNet call (a unit's SLOT - its place in the exchange buffer - stays its index in
        # `mine`).  context_batch_size 1: the two branches of a window go together, [uncond, cond] - the reference's batch
        # (:759-763), and the one whose halves share their prefix (unet._begin); units whose sibling lives on another rank are
        # batched in pairs.  context_batch_size > 1: consecutive units, 2 * cbs per call.
        if cbs == 1 and cfg:
            byw = {}
            for k, (w_, br) in enumerate(mine):
                byw.setdefault(w_, {})[br] = k
            chunks = [[byw[w_][0], byw[w_][1]] for w_ in sorted(byw) if len(byw[w_]) == 2]
            singles = sorted((k for w_ in byw if len(byw[w_]) == 1 for k in byw[w_].values()), key=lambda k: (mine[k][1], k))
            chunks += [singles[i:i + 2] for i in range(0, len(singles), 2)]
        else:
            n_per = len(st.branches) * cbs
            chunks = [list(range(i, min(i + n_per, len(mine)))) for i in range(0, len(mine), n_per)]
        for idxs in chunks:
            chunk = [mine[k] for k in idxs]
            uc = [k for k, u in enumerate(chunk) if u[1] == 0]
            # one UNet call reads ONE bank (the row named by a device word): cond units are batched per bank variant
            for n_call, tv in enumerate(sorted({st.unit_tv[u] for u in chunk if u[1] == 1}, reverse=True) or [None]):
                order = (uc if n_call == 0 else []) + [k for k, u in enumerate(chunk) if u[1] == 1 and st.unit_tv[u] == tv]
                call = SimpleNamespace(units=[chunk[k] for k in order], slots=[idxs[k] for k in order], bank_tv=tv)
                call.n_uc = sum(1 for u in call.units if u[1] == 0)
                # [uncond windows..., cond windows...] over the SAME windows in the same order: the two halves of the UNet batch
                # carry identical latents and differ only from the first cross-attention on (unet._begin shares that prefix)
                call.halves_identical = (call.n_uc * 2 == len(call.units) and call.n_uc > 0 and
                                         [u[0] for u in call.units[:call.n_uc]] == [u[0] for u in call.units[call.n_uc:]])
                call.idx = [torch.tensor(st.windows[w], dtype=torch.int64, device=dev) for w, _ in call.units]
                st.calls.append(call)
        if st.emulate:          # no exchange: the accumulators see this rank's units only
            st.units = mine
        st.t_table = torch.tensor(st.timesteps, dtype=torch.int64, device=dev)   # INT timestep table, bit-exact
        st.t_buf = torch.zeros(1, dtype=torch.int64, device=dev)
        if use_graphs is None:      # the measured path is the default one on a HIP device
            use_graphs = dev.type == "cuda"
        st.use_graphs, st.graphs = bool(use_graphs) and dev.type == "cuda", {}
        st.graph_pool = st.writer_pool = None
        if st.use_graphs:
            st.graph_pool = torch.cuda.graph_pool_handle()
            st.writer_pool = torch.cuda.graph_pool_handle()   # own pool: the ReferenceNet pass runs concurrently with the Backbone
        # ---- contexts of the attn2 layers: their K / V^T projections depend on the context only -> once per clip (_bind_inputs)
        st.audio_features = None if audio_features is None else torch.empty(tuple(audio_features.shape), device=dev, dtype=torch.float32)
        if speed_embeddings is not None and speed_embeddings.shape[0] not in (1, 2):
            raise ValueError("speed_embeddings must have 1 row (shared) or 2 rows [uncond, cond]")
        st.speed = None if speed_embeddings is None else torch.empty(tuple(speed_embeddings.shape), device=dev, dtype=torch.float32)
        for call in st.calls:
            call.ctx = call.ctx_kv = call.speed = None
        st.text_c = st.text[1:2]
        st.ref_ctx_kv = {}
        # ---- eps hand-off: every unit's rows (nf*HW, C4); slot = position in the rank's unit list
        st.send = torch.zeros(st.n_slots, nf * st.HW, st.C4, device=dev, dtype=unet.dtype)
        st.recv = torch.zeros(st.world_size, st.n_slots, nf * st.HW, st.C4, device=dev, dtype=unet.dtype) if st.dist else None
        st.noise_pred = torch.empty(len(st.branches), st.C4, st.f_tot, st.HW, device=dev, dtype=torch.float32)
        st.counter = torch.empty(st.f_tot, device=dev, dtype=torch.float32)
        st.return_eps, st.eps_trace = return_eps, []
        # ---- ReferenceNet groups: the banks depend on the timestep only (never on the latents): T timesteps per pass
        st.T = T = self.reference_group_size(reference_group, n_steps, st.world_size if (st.dist or st.emulate) else 1)
        st.groups = [list(range(i, min(i + T, n_steps))) for i in range(0, n_steps, T)]
        st.ref_t = torch.zeros(T, dtype=torch.int64, device=dev)
        st.row_table = torch.tensor([((s_ // T) % 2) * T + s_ % T for s_ in range(n_steps)], dtype=torch.int32, device=dev)
        st.bank_idx = torch.zeros(1, dtype=torch.int32, device=dev)
        st.kv_all, st.group_ready, st.group_pending, st.groups_launched = {}, -1, -1, 0
        st.ref_sel, st.ref_recv, st.ref_gath = {}, {}, {}
        st.bank_pack, st.stage = {}, {}
        spec_r = appearance_encoder.spec
        chan = {a.prefix: a.channels for blk in spec_r.down + [spec_r.mid] + spec_r.up for a in blk.attentions if a is not None}
        st.bank_C = [chan[p] for p in st.writer.order]
        # look-ahead = group g+1's ReferenceNet pass on a second stream under group g's Backbone steps.  With world_size > 1 the
        # side stream carries the WRITE passes only; the bank all_gather stays on the main stream and on the communicator of the
        # per-step eps all_gather (see _ensure_group): one program order of collectives on every rank, so the overlap is safe
        # at any world size and on by default.
        if reference_lookahead is None:
            reference_lookahead = True
        st.lookahead = bool(reference_lookahead) and dev.type == "cuda"
        st.side = torch.cuda.Stream() if st.lookahead else None
        # ControlNet branch (EMOAnimationPipeline.py:643-650,678-679,718-746): (F_tot,3,H,W) conditioning images in [0,1]
        st.controlnet = controlnet
        if controlnet is not None:
            if controlnet_cond is None or controlnet_cond.shape[0] != st.f_tot:
                raise ValueError("controlnet_cond must hold one (3,H,W) conditioning image per frame")
            st.cn_cond = torch.empty(tuple(controlnet_cond.shape), device=dev, dtype=torch.float32)
            # the frames this rank's units touch; residuals are computed once per frame and step, in chunks of
            # context_frames (the reference caches them per frame from overlap-0 windows: the network is per-frame, so
            # the chunking does not enter the result)
            need = sorted({k for call in st.calls for w, _ in call.units for k in st.windows[w]})
            st.cn_pos = {k: i for i, k in enumerate(need)}
            st.cn_chunks = [torch.tensor(need[i:i + context_frames], dtype=torch.int64, device=dev) for i in range(0, len(need), context_frames)]
            st.cn_text = st.text_c                                   # cond text embedding (:678-679)
            for call in st.calls:
                call.cn_sel = torch.tensor([st.cn_pos[k] for w, _ in call.units for k in st.windows[w]], dtype=torch.int64, device=dev)
            st.cn_down, st.cn_mid, st.cn_embed = None, None, None
        self._bind_inputs(st, latents, ref_image_latents, text_embeddings, audio_features=audio_features, speed_embeddings=speed_embeddings,
                          motion_latents=motion_latents, controlnet_cond=controlnet_cond,
                          controlnet_conditioning_scale=controlnet_conditioning_scale, guidance_scale=guidance_scale, eta=eta, seed=seed)
        return st

    @staticmethod
    def reference_group_size(reference_group, n_steps, world_size=1):
        """ReferenceNet timesteps per batched pass: at least 1, never more than the loop has steps; with world_size > 1 the ranks
        deal a group's timesteps among themselves, and a size that is not a multiple of world_size would pad every rank to
        ceil(T / world) timesteps (10 over 8 ranks: 16 computed and shipped, 10 used) - rounded up to a multiple instead, still
        capped by the step count."""
        T = max(1, min(int(reference_group), int(n_steps)))
        if world_size > 1:
            T = min(-(-T // world_size) * world_size, int(n_steps))
        return T

    # ---- the INPUTS of a prepared loop state, copied into its buffers in place (the captured graphs keep reading them)
    @staticmethod
    def _do_cfg(guidance_scale):
        """`do_classifier_free_guidance = guidance_scale > 1.0` (:622), decided on the f32 value emo_cfg_step receives: a scale
        in (1, 1 + 2^-24] would otherwise allocate two noise_pred planes for a kernel that reads one."""
        import numpy as np
        return bool(np.float32(guidance_scale) > np.float32(1.0))

    @staticmethod
    def _text_rows(text_embeddings, cfg):
        if text_embeddings.shape[0] == 2 and not cfg:
            text_embeddings = text_embeddings[1:]          # [uncond, cond] handed over anyway: the cond row is the prompt
        if text_embeddings.shape[0] != (2 if cfg else 1):
            raise ValueError("text_embeddings must be (2, L, D) = [uncond, cond] (:631), or (1, L, D) = [cond] without guidance")
        return text_embeddings

    @staticmethod
    def _motion_rows(motion_latents):
        ml = motion_latents
        if ml.dim() == 5:
            ml = ml[0].permute(1, 0, 2, 3)
        return ml

    def _bind_inputs(self, st, latents, ref_image_latents=None, text_embeddings=None, *, audio_features=None, speed_embeddings=None,
                     motion_latents=None, controlnet_cond=None, controlnet_conditioning_scale=None, guidance_scale=None, eta=None,
                     seed=None):
        """Copy a clip's inputs into the buffers of a prepared state (None = keep what is bound).  Everything derived from them -
        the attn2 K / V^T of the text / audio context per UNet call, the ReferenceNet's text K / V^T - is recomputed INTO the
        tensors the captured graphs already read; the ReferenceNet groups are marked stale (the reference image and the motion
        frames enter them)."""
        unet, dev = self.unet, st.latents.device

        def put(dst, src, what):
            src = src.to(dev).float()
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"{what} {tuple(src.shape)} != prepared {tuple(dst.shape)}")
            dst.copy_(src)
        if st.side is not None:   # nothing of the previous clip may still be writing the bank cache
            torch.cuda.current_stream().wait_stream(st.side)
        put(st.latents, latents, "latents")
        if ref_image_latents is not None:
            put(st.ref_lat[:1], ref_image_latents.reshape(1, st.C4, st.h, st.w), "ref_image_latents")
        if motion_latents is not None:
            ml = self._motion_rows(motion_latents)
            if st.n_ref_images == 1 or tuple(ml.shape) != (st.n_ref_images - 1, st.C4, st.h, st.w):
                raise ValueError(f"motion_latents must be ({st.n_ref_images - 1}, {st.C4}, {st.h}, {st.w}), got {tuple(ml.shape)}")
            put(st.ref_lat[1:], ml, "motion_latents")
        elif st.n_ref_images != 1 and ref_image_latents is not None:
            raise ValueError("the state was prepared with motion frames: pass motion_latents")
        if guidance_scale is not None:
            if self._do_cfg(guidance_scale) != st.cfg:
                raise ValueError("guidance_scale crosses 1.0: classifier-free guidance on / off is part of the plan (prepare again)")
            st.guidance_scale = float(guidance_scale)
        if eta is not None:
            st.eta = eta
        if seed is not None:
            st.seed = seed
        ctx_stale = False
        if text_embeddings is not None:
            t = self._text_rows(text_embeddings, st.cfg).to(dev).float()
            put(st.text, t if st.cfg else t.expand(2, -1, -1), "text_embeddings")
            ctx_stale = True
            for tv in st.bank_variants:
                for pr, kv in st.appearance_encoder.context_kv(st.text[tv:tv + 1]).items():
                    self._put_kv(st.ref_ctx_kv.setdefault(tv, {}), pr, kv)
        if audio_features is not None:
            if st.audio_features is None:
                raise ValueError("the state was prepared without audio_features")
            put(st.audio_features, audio_features, "audio_features")
            ctx_stale = True
        if speed_embeddings is not None:
            if st.speed is None:
                raise ValueError("the state was prepared without speed_embeddings")
            put(st.speed, speed_embeddings, "speed_embeddings")
        for call in st.calls:
            if ctx_stale:
                if st.audio_features is None:
                    ctx = torch.cat([st.text[st.unit_tv[u]:st.unit_tv[u] + 1] for u in call.units])    # (n, L, D)
                else:   # per-frame audio context; uncond units get a zero context (design choice)
                    parts = []
                    for (w, br), ix in zip(call.units, call.idx):
                        a = st.audio_features.index_select(0, ix)
                        parts.append(a if br == 1 else torch.zeros_like(a))
                    ctx = torch.cat(parts)                                                              # (n*F, L_a, D)
                if call.ctx is None:
                    call.ctx = ctx
                else:
                    call.ctx.copy_(ctx)
                if call.ctx_kv is None:
                    call.ctx_kv = {}
                for pr, kv in unet.context_kv(call.ctx).items():
                    self._put_kv(call.ctx_kv, pr, kv)
            if speed_embeddings is not None:
                se = st.speed
                sp = torch.cat([se[(br if se.shape[0] == 2 else 0):(br if se.shape[0] == 2 else 0) + 1] for _, br in call.units])
                if call.speed is None:
                    call.speed = sp
                else:
                    call.speed.copy_(sp)
        if controlnet_cond is not None:
            if st.controlnet is None:
                raise ValueError("the state was prepared without a ControlNet")
            put(st.cn_cond, controlnet_cond, "controlnet_cond")
            # the conditioning embedding (controlnet.py:523) depends on the conditioning images only: once per clip, not per step
            embeds = [st.controlnet.cond_embedding(st.cn_cond.index_select(0, idx)) for idx in st.cn_chunks]
            if getattr(st, "cn_embed", None) is None:
                st.cn_embed = embeds
            else:
                for (dst, _, _), (src, _, _) in zip(st.cn_embed, embeds):
                    dst.copy_(src)
        if controlnet_conditioning_scale is not None and st.controlnet is not None:
            scale = float(controlnet_conditioning_scale)
            if getattr(st, "cn_scale", scale) != scale and st.graphs.get("controlnet") not in (None, "warm"):
                raise ValueError("controlnet_conditioning_scale is baked into the captured ControlNet pass (prepare again)")
            st.cn_scale = scale
        st.group_ready, st.group_pending, st.eps_trace = -1, -1, []

    @staticmethod
    def _put_kv(store, key, kv):
        """store[key] = (K rows, V^T) - into the tensors already there, if any (captured graphs hold their addresses)"""
        if key in store:
            for dst, src in zip((store[key],) if torch.is_tensor(store[key]) else store[key], (kv,) if torch.is_tensor(kv) else kv):
                dst.copy_(src)
        else:
            store[key] = kv

    # ---- ReferenceNet: one batched write pass per GROUP of timesteps, one group ahead of the Backbone on a second stream
    def _ref_sel(self, st, Tg):
        """group-local timestep indices this rank computes (rank r: r, r+world, ...; padded by repeating the last one)"""
        if Tg not in st.ref_sel:
            n = -(-Tg // st.world_size)
            mine = [min(st.rank + i * st.world_size, Tg - 1) for i in range(n)]
            st.ref_sel[Tg] = torch.tensor(mine, dtype=torch.int64, device=st.ref_t.device)
        return st.ref_sel[Tg]

    def _part_reference_write(self, st, Tg, tv):
        """ReferenceNet write pass (:711-716) under text variant tv (1 = the cond text) for this rank's share of the group's
        timesteps, batched: (n,4,h,w) with one timestep per row.  The reference runs [uncond-text, cond-text] copies every
        step, but at context_batch_size 1 the reader never uses the uncond bank row (for the uc rows `hidden_states_c` is
        overwritten by the bank-free uc attention, mutual_self_attention.py:243-256) and the ReferenceNet is batch-independent:
        only the variants some cond unit reads (st.bank_variants) are computed.  Banks are rounded through fp16 like
        reader.update() does (:588) and packed for the exchange."""
        t = st.ref_t[:Tg] if st.world_size == 1 else st.ref_t.index_select(0, self._ref_sel(st, Tg))
        n, k = t.numel(), st.n_ref_images
        st.appearance_encoder._reference_control = st.writer   # (another prepared state on the same models may have attached its own)
        st.writer.clear()
        # batch rows timestep-major: (t0: reference image, motion frames...), (t1: ...) - the k images of one timestep are
        # consecutive, so their LN1 rows (k, L, C) read as ONE bank of k*L tokens (the reference concatenates several bank
        # entries along tokens the same way, mutual_self_attention.py:239)
        imgs = st.ref_lat.expand(n, -1, -1, -1) if k == 1 else st.ref_lat.repeat(n, 1, 1, 1)
        tt = t if k == 1 else t.repeat_interleave(k)
        st.appearance_encoder(imgs, tt, encoder_hidden_states=st.text[tv:tv + 1], return_dict=False, _ctx_kv=st.ref_ctx_kv[tv])
        tgt = self.unet.dtype
        st.bank_L = [k * st.writer.bank[p][0].shape[1] for p in st.writer.order]
        banks = [ops.convert(st.writer.bank[p][0], tgt, fp16_round=True).reshape(n, -1) for p in st.writer.order]
        st.writer.clear()                                                                      # :823
        st.bank_pack[tv] = banks if st.world_size == 1 else torch.cat(banks, dim=1)            # (n, total) plumbing copy

    def _part_reference_project(self, st, Tg, tv):
        """K / V^T projections of the group's banks with the BACKBONE's attn1.to_k / to_v (the read side of
        mutual_self_attention.py:238-241), once per group instead of once per step."""
        st.stage[tv] = {}
        off = 0
        for i, (pr, L, C_) in enumerate(zip(st.reader.order, st.bank_L, st.bank_C)):
            if st.world_size == 1:
                rows = st.bank_pack[tv][i]
            else:   # bank i occupies columns [off, off + L*C) of every gathered row (rows in group order)
                rows = st.ref_gath[(Tg, tv)][:Tg, off:off + L * C_].contiguous()
                off += L * C_
            k, vt = self.unet.bank_kv(pr, rows.reshape(-1, C_), L)
            st.stage[tv][pr] = (k, vt, L)

    def _reference_write(self, st, g):
        """Phase 1 of group g: this rank's share of the group's ReferenceNet write passes (no communication) on the CURRENT stream."""
        steps = st.groups[g]
        Tg = len(steps)
        st.groups_launched += 1
        st.ref_t[:Tg].copy_(st.t_table[steps[0]:steps[0] + Tg], non_blocking=True)
        for tv in st.bank_variants:
            self._run(st, ("ref_write", Tg, tv), lambda tv=tv: self._part_reference_write(st, Tg, tv), pool=st.writer_pool)

    def _reference_finish(self, st, g):
        """Phase 2 of group g on the CURRENT stream: the bank exchange (world_size > 1), the K / V^T projection of the whole group
        with the Backbone's weights, and the copy into slot g % 2 of the resident cache."""
        steps = st.groups[g]
        Tg, T, slot = len(steps), st.T, g % 2
        for tv in st.bank_variants:
            if st.world_size > 1:
                # north_star: "RCCL all-gather over xGMI to broadcast ReferenceNet features" - rank r computed timesteps
                # r, r+world, ... of the group; ONE all_gather hands every rank every timestep's banks.  Same communicator and same
                # stream as the per-step eps all_gather: every rank issues its collectives in ONE program order (eps of the group's
                # last step, banks of the next group, eps of its first step, ...), so no start-order hazard exists by construction
                import torch.distributed as td
                send = st.bank_pack[tv]
                n = send.shape[0]
                if (Tg, tv) not in st.ref_gath:
                    st.ref_recv[(Tg, tv)] = torch.empty(st.world_size * n, send.shape[1], device=send.device, dtype=send.dtype)
                    st.ref_gath[(Tg, tv)] = torch.empty(st.world_size * n, send.shape[1], device=send.device, dtype=send.dtype)
                if st.emulate:      # (stands in for the gathered rows: this rank's banks, repeated)
                    st.ref_recv[(Tg, tv)].view(st.world_size, -1).copy_(send.contiguous().view(1, -1).expand(st.world_size, -1))
                else:
                    td.all_gather_into_tensor(st.ref_recv[(Tg, tv)].view(-1), send.contiguous().view(-1))
                # row (rank r, slot i) is group-local timestep i*world + r -> group order
                st.ref_gath[(Tg, tv)].view(n, st.world_size, -1).copy_(st.ref_recv[(Tg, tv)].view(st.world_size, n, -1).transpose(0, 1))
            self._run(st, ("ref_project", Tg, tv), lambda tv=tv: self._part_reference_project(st, Tg, tv), pool=st.writer_pool)
            if tv not in st.kv_all:   # resident cache: two groups of T timesteps per block
                st.kv_all[tv] = {}
                for pr, (k, vt, L) in st.stage[tv].items():
                    st.kv_all[tv][pr] = (torch.zeros(2 * T * L, k.shape[1], device=k.device, dtype=k.dtype),
                                         torch.zeros(2 * T, vt.shape[1], vt.shape[2], device=vt.device, dtype=vt.dtype), L)
            dsts, srcs = [], []
            for pr, (k, vt, L) in st.stage[tv].items():
                ka, va, _ = st.kv_all[tv][pr]
                dsts += [ka[slot * T * L:slot * T * L + Tg * L], va[slot * T:slot * T + Tg]]
                srcs += [k[:Tg * L], vt[:Tg]]
            torch._foreach_copy_(dsts, srcs)

    def _reference_group(self, st, g):
        """Compute group g's projected banks on the CURRENT stream and store them in slot g % 2 of the resident cache."""
        self._reference_write(st, g)
        self._reference_finish(st, g)

    def _ensure_group(self, st, si):
        """Group of step si resident (waiting for / computing it if needed), the NEXT group launched on the side stream.
        world_size 1: the whole pass of group g+1 (write, projection, cache copy) rides the side stream.  world_size > 1: only its
        WRITE pass does (the ReferenceNet forwards - the expensive, communication-free part); the bank all_gather and the projection are
        issued on the MAIN stream when the loop reaches the group - in program order with the per-step eps all_gather, on the one
        communicator both use (EMOAnimationPipeline.py:796-821 is the exchange this replaces)."""
        g = si // st.T
        cuda = self.unet.device.type == "cuda"
        main = torch.cuda.current_stream() if cuda else None
        split = st.world_size > 1
        if st.group_ready != g:
            if st.group_pending == g:
                main.wait_stream(st.side)
                if split:
                    self._reference_finish(st, g)
            else:
                if st.group_pending >= 0 and cuda:   # a look-ahead pass for ANOTHER group is in flight (the caller jumped): it shares the
                    main.wait_stream(st.side)        # timestep buffer and the captured write graphs with the pass below
                self._reference_group(st, g)
            st.group_ready, st.group_pending = g, -1
            if g + 1 < len(st.groups) and st.lookahead:
                # slot (g+1) % 2 held group g-1: every step of it is already enqueued on the main stream
                st.side.wait_stream(main)
                with torch.cuda.stream(st.side):
                    if split:
                        self._reference_write(st, g + 1)
                    else:
                        self._reference_group(st, g + 1)
                st.group_pending = g + 1

    # ---- ControlNet (per-frame residual cache of the step, :718-746)
    def _part_controlnet(self, st):
        downs, mids = [], []
        for i, idx in enumerate(st.cn_chunks):
            x = self.scheduler.scale_model_input(st.latents.index_select(2, idx), None)[0].permute(1, 0, 2, 3).contiguous()
            d, m = st.controlnet(x, st.t_buf, encoder_hidden_states=st.cn_text.repeat(idx.numel(), 1, 1), controlnet_cond=None,
                                 conditioning_scale=st.cn_scale, return_dict=False, _cond_rows=st.cn_embed[i])
            downs.append(d)
            mids.append(m)
        st.cn_down = [torch.cat([d[i] for d in downs]) for i in range(len(downs[0]))]
        st.cn_mid = torch.cat(mids)

    def _controlnet_residuals(self, st, call):
        """select_controlnet_res_samples (:514-540): frames of the call's units, '(b f) c h w -> b c f h w' (the CFG repeat of
        the reference is the uncond and the cond unit of a window selecting the same frames)."""
        if st.controlnet is None:
            return {}
        n, nf = len(call.units), st.nf

        def to5(t):
            y = t.index_select(0, call.cn_sel)
            return y.reshape(n, nf, *y.shape[1:]).permute(0, 2, 1, 3, 4)
        return dict(down_block_additional_residuals=tuple(to5(t) for t in st.cn_down), mid_block_additional_residual=to5(st.cn_mid))

    # ---- one UNet call = the rank's next <= 2*cbs units; reads the timestep from the device buffer st.t_buf and the bank row
    #      from st.bank_idx, so it is captured once into a HIP graph and replayed for the other steps
    def _part_unet(self, st, ci):
        call = st.calls[ci]
        x = torch.cat([st.latents.index_select(2, ix) for ix in call.idx])                     # :759-763 (index/copy only)
        x = self.scheduler.scale_model_input(x, None)
        # (a call of uncond units only names any resident cache: every one of its batches skips the bank segment)
        bank_tv = call.bank_tv if call.bank_tv is not None else st.bank_variants[0]
        self.unet._reference_control = st.reader
        st.reader.set_projected_banks(st.kv_all[bank_tv], st.bank_idx, call.n_uc)               # replaces reader.update (:774)
        rows = self.unet(x, st.t_buf, encoder_hidden_states=call.ctx, speed_embeddings=call.speed, return_dict=False,
                         _return_rows=True, _ctx_kv=call.ctx_kv, _halves_identical=call.halves_identical,
                         **self._controlnet_residuals(st, call))   # :777-786
        n_rows = st.nf * st.HW
        for k, slot in enumerate(call.slots):
            st.send[slot].copy_(rows[k * n_rows:(k + 1) * n_rows])

    def _accumulate_all(self, st):
        """noise_pred[:, :, c] += pred; counter[:, :, c] += 1 (:790-794) over EVERY unit in global order - each rank does this
        redundantly on the gathered rows, so the accumulators (and the latents) are bit-identical on all ranks."""
        for i, (w, br) in enumerate(st.units):
            rows = st.send[i] if not st.dist else st.recv[i % st.world_size, i // st.world_size]
            ops.accumulate_window(rows, st.noise_pred[st.np_slot[br]], st.counter, st.frame_idx[w], C_=st.C4, F=st.f_tot, HW=st.HW,
                                  add_counter=(br == st.branches[0]))

    def _run(self, st, key, fn, pool=None):
        """Eager on first use (warm-up: lazy kernel attributes, allocator), HIP-graph capture on the second, replay after.
        Launch-bound host code (~1700 kernel launches per step from Python) disappears from the critical path."""
        if not st.use_graphs or ops.PROFILER is not None:
            return fn()
        state = st.graphs.get(key)
        if state is None:
            fn()
            st.graphs[key] = "warm"
            return
        if state == "warm":
            g = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are checked during capture - the RCCL watchdog thread of a multi-GPU
            # run polls events concurrently and must not invalidate the capture
            with torch.cuda.graph(g, pool=pool if pool is not None else st.graph_pool, capture_error_mode="thread_local"):
                fn()
            st.graphs[key] = g
            state = g
        state.replay()

    @torch.no_grad()
    def denoise_step(self, st, si):
        """One iteration of the hot loop (EMOAnimationPipeline.py:698-823)."""
        sch = self.scheduler
        dev = self.unet.device
        t = st.timesteps[si]
        self._ensure_group(st, si)                                     # :711-716, hoisted: banks of T timesteps per pass
        st.t_buf.copy_(st.t_table[si:si + 1], non_blocking=True)       # device-to-device: the INT timestep of this step
        st.bank_idx.copy_(st.row_table[si:si + 1], non_blocking=True)  # ... and its row in the resident bank cache
        st.noise_pred.zero_()
        st.counter.zero_()
        if st.emulate:
            st.counter.add_(1e-6)       # frames of the other ranks' windows: 0 / 1e-6 instead of 0 / 0
        if st.controlnet is not None and st.calls:   # per-frame residual cache of this step (:718-746)
            self._run(st, "controlnet", lambda: self._part_controlnet(st))
        for ci in range(len(st.calls)):                                                        # :757
            self._run(st, ("unet", ci), lambda ci=ci: self._part_unet(st, ci))
        if st.dist:   # replaces gather-to-root (:796-809) + broadcast (:819-821): ONE all_gather of the eps slices
            import torch.distributed as td
            td.all_gather_into_tensor(st.recv.view(-1), st.send.view(-1))
        self._accumulate_all(st)
        # (the state's own scheduler object and step count: `self.scheduler` may have been replaced or re-timed since the plan was made)
        sch = st.scheduler
        c_x, c_eps, c_n = self._coefficients(sch, t, st.eta if isinstance(sch, DDIMScheduler) else None, st.num_inference_steps)
        eps_out = torch.empty(st.C4 * st.f_tot * st.HW, device=dev, dtype=torch.float32) if st.return_eps else None
        ops.cfg_step(st.noise_pred, st.counter, st.latents, C_=st.C4, F=st.f_tot, HW=st.HW, guidance_scale=st.guidance_scale,
                     c_x=c_x, c_eps=c_eps, c_noise=c_n, seed=st.seed, step=si, eps_out=eps_out)  # :812-817 fused
        if st.return_eps:
            st.eps_trace.append(eps_out.view(1, st.C4, st.f_tot, st.h, st.w))

    @staticmethod
    def _coefficients(sch, t, eta, num_inference_steps):
        """(c_x, c_eps, c_noise) of step t from the scheduler object.  The step count goes by KEYWORD, and a user-supplied scheduler
        with the leaner `coefficients(t[, eta])` signature is still served - as long as it has been timed for this many steps."""
        try:
            return sch.coefficients(t, eta, num_inference_steps=num_inference_steps)
        except TypeError:
            if getattr(sch, "num_inference_steps", num_inference_steps) != num_inference_steps:
                raise
            return sch.coefficients(t, eta) if eta is not None else sch.coefficients(t)

    def _run_loop(self, st, num_actual_inference_steps=None, callback=None, callback_steps=1):
        n = st.num_inference_steps
        for si, t in enumerate(st.timesteps):
            if num_actual_inference_steps is not None and si < n - num_actual_inference_steps:   # :699-700
                continue
            self.denoise_step(st, si)
            if callback is not None and si % callback_steps == 0:
                callback(si, t, st.latents)
        return (st.latents, st.eps_trace) if st.return_eps else st.latents

    @torch.no_grad()
    def denoise(self, latents, ref_image_latents, text_embeddings, *, num_actual_inference_steps=None, callback=None,
                callback_steps=1, reuse_state=False, **kw):
        """The whole loop; returns the denoised latents f32 (1,4,F_tot,h,w) (and the eps trace if asked).
        reuse_state: keep the prepared state (plan + captured HIP graphs) on the pipeline and re-arm it when the next call has the
        same plan - same shapes, step count, window / guidance / distribution settings and the same loaded weights; only the
        inputs are copied in (`_bind_inputs`).  `__call__` turns it on: a second clip costs the loop and nothing else."""
        if not reuse_state:
            st = self.prepare_denoise(latents, ref_image_latents, text_embeddings, **kw)
            return self._run_loop(st, num_actual_inference_steps, callback, callback_steps)
        key = self._plan_key(latents, ref_image_latents, text_embeddings, kw)
        cached = self._plan_cache
        if cached is not None and cached[0] == key:
            st = cached[1]
            bind = {k: kw[k] for k in ("audio_features", "speed_embeddings", "motion_latents", "controlnet_cond",
                                       "controlnet_conditioning_scale") if kw.get(k) is not None}
            # an omitted guidance_scale / eta / seed means the documented DEFAULT, not "what the previous clip used" (they are not
            # part of the plan key, so the hit happens either way; "None = keep" is reset_denoise's contract, not this one's)
            bind.update(guidance_scale=_default(kw.get("guidance_scale"), 7.5), eta=_default(kw.get("eta"), 0.0), seed=_default(kw.get("seed"), 0))
            self._bind_inputs(st, latents, ref_image_latents, text_embeddings, **bind)
        else:
            self._plan_cache = None          # drop the old graphs before the new plan allocates
            st = self.prepare_denoise(latents, ref_image_latents, text_embeddings, **kw)
            self._plan_cache = (key, st)
        res = self._run_loop(st, num_actual_inference_steps, callback, callback_steps)
        # (the state's latents are overwritten by the next clip)
        return (res[0].clone(), list(res[1])) if isinstance(res, tuple) else res.clone()

    def clear_plan_cache(self):
        """Release the prepared state `denoise(reuse_state=True)` / `__call__` keep (buffers, captured HIP graphs, bank cache)."""
        self._plan_cache = None

    def _plan_key(self, latents, ref_image_latents, text_embeddings, kw):
        """Everything a prepared state's PLAN depends on (prepare_denoise above `_bind_inputs`): tensor shapes, the settings that
        shape windows / units / groups / graphs, and the identity of the packed weights the graphs were captured over."""
        from . import unet as unet_mod

        def shp(t):
            return None if t is None else tuple(t.shape)

        def wid(m):   # a re-pack (load_state_dict / .to) builds a new dict of packed tensors: captured graphs are stale
            return None if m is None else (id(m), id(getattr(m, "_w", None)), str(getattr(m, "dtype", None)))
        def sched_id(s):   # the scheduler OBJECT the state steps with and everything its tables depend on
            c = s.config   # (a user-supplied scheduler may carry a leaner config: absent fields key as None)
            return (id(s), type(s).__name__) + tuple(getattr(c, f, None) for f in ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule",
                                                                                   "steps_offset", "set_alpha_to_one", "timestep_spacing"))

        def pg_id():       # a state prepared with dist=True holds communicators of the default group it was made under
            if not kw.get("dist"):
                return None
            import torch.distributed as td
            return id(td.group.WORLD) if td.is_initialized() else None     # (the public handle of the default group)
        plan = {k: v for k, v in kw.items() if k not in ("audio_features", "speed_embeddings", "motion_latents", "controlnet_cond",
                                                            "controlnet_conditioning_scale", "guidance_scale", "eta", "seed",
                                                            "appearance_encoder", "controlnet")}
        return (shp(latents), shp(ref_image_latents), shp(text_embeddings), shp(kw.get("audio_features")), shp(kw.get("speed_embeddings")),
                shp(kw.get("motion_latents")), shp(kw.get("controlnet_cond")), kw.get("controlnet_conditioning_scale", 1.0),
                self._do_cfg(kw.get("guidance_scale", 7.5)), wid(self.unet), wid(kw.get("appearance_encoder")), wid(kw.get("controlnet")),
                sched_id(self.scheduler), pg_id(), unet_mod.SHARE_CFG_PREFIX, unet_mod.GN_FOLD_MIN_HW, unet_mod.GN_CONV_MIN_HW, unet_mod.MERGE_QKV,
                tuple(sorted((k, repr(v)) for k, v in plan.items())))

    def reset_denoise(self, st, latents, motion_latents=None, **inputs):
        """Re-arm a prepared loop state for another clip of the SAME geometry: new noisy latents (and motion frames; optionally
        ref_image_latents / text_embeddings / audio_features / speed_embeddings / controlnet_cond / guidance_scale / eta / seed)
        are copied IN PLACE, so the captured HIP graphs, the resident bank cache and the communicators are reused - only the
        context K / V^T and the ReferenceNet groups are recomputed (the reference image and the motion frames enter them)."""
        if tuple(latents.shape) != tuple(st.latents.shape):
            raise ValueError(f"reset_denoise: latents {tuple(latents.shape)} != prepared {tuple(st.latents.shape)}")
        if motion_latents is None and st.n_ref_images != 1:
            raise ValueError("reset_denoise: the state was prepared with motion frames")
        if motion_latents is not None:
            ml = self._motion_rows(motion_latents)
            if ml.shape[0] != st.n_ref_images - 1 or tuple(ml.shape[1:]) != (st.C4, st.h, st.w):
                raise ValueError(f"reset_denoise: motion_latents must be ({st.n_ref_images - 1}, {st.C4}, {st.h}, {st.w}), got {tuple(ml.shape)}")
        self._bind_inputs(st, latents, motion_latents=motion_latents, **inputs)
        return st

    @torch.no_grad()
    def denoise_chained(self, clips, ref_image_latents, text_embeddings, *, n_motion_frames=4, **kw):
        """Long-video generation by clip chaining (intent of Net.py:56-72 `pre_extract_motion_features` and
        junk/EMo-write-up.txt:104-110 - the reference never wires it): clip k+1 is denoised with the LAST n_motion_frames
        latent frames of the denoised clip k as motion frames; the first clip gets zero maps.  The motion frames pass through
        the ReferenceNet together with the reference image at every timestep and their LN1 features are appended to the
        reference banks as extra tokens, i.e. every spatial self-attention of the Backbone's mid / up path also attends to
        the previous clip's tail (same bank mechanism, Lk1 = (1 + n_motion_frames) * L).  `clips`: list of noisy latents
        (1, 4, F, h, w), each with F >= n_motion_frames.  Clips of one geometry share ONE prepared state (graphs, bank cache,
        context K/V); a clip of another shape prepares afresh.  Returns the list of denoised latents.  Design choice - no
        reference behaviour to match."""
        out, prev, st = [], None, None
        run_kw = {k: kw.pop(k) for k in ("num_actual_inference_steps", "callback", "callback_steps") if k in kw}
        for lat in clips:
            _, c4, f, h, w = lat.shape
            if n_motion_frames > 0:
                if f < n_motion_frames:
                    raise ValueError(f"denoise_chained: a clip of {f} frames cannot hand over {n_motion_frames} motion frames")
                motion = torch.zeros(n_motion_frames, c4, h, w) if prev is None else prev[0, :, -n_motion_frames:].permute(1, 0, 2, 3)
                if motion.shape[0] != n_motion_frames or tuple(motion.shape[2:]) != (h, w):
                    raise ValueError("denoise_chained: consecutive clips must share the latent size")
            else:
                motion = None
            if st is not None and tuple(st.latents.shape) == tuple(lat.shape):
                self.reset_denoise(st, lat, motion)
            else:
                st = self.prepare_denoise(lat, ref_image_latents, text_embeddings, motion_latents=motion, **kw)
            res = self._run_loop(st, **run_kw)
            prev = (res[0] if isinstance(res, tuple) else res).clone()   # (the state's latents are overwritten by the next clip)
            out.append(prev)
        return out

    @torch.no_grad()
    def images2latents(self, images, dtype=None):
        """EMOAnimationPipeline.py:402-414: uint8 RGB frames (f, h, w, c) -> `/ 127.5 - 1`, 'f h w c -> f c h w',
        `vae.encode(frame)['latent_dist'].mean * 0.18215`, frame by frame."""
        if self.vae is None:
            raise ValueError("images2latents needs a VAE (emote_hack_amd.vae.AutoencoderKL)")
        import numpy as np
        x = torch.from_numpy(np.ascontiguousarray(images)) if isinstance(images, np.ndarray) else torch.as_tensor(images)
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError(f"images2latents: expected (f, h, w, 3) RGB frames, got {tuple(x.shape)}")
        x = (x.float() / 127.5 - 1.0).permute(0, 3, 1, 2).contiguous()
        return torch.cat([self.vae.encode(x[i:i + 1])["latent_dist"].mean * 0.18215 for i in range(x.shape[0])])

    def _source_image_latents(self, source_image, width, height):
        """:686-689: a path is opened and resized to (width, height); an (H, W, 3) uint8 array is taken as it is."""
        import numpy as np
        if isinstance(source_image, str):
            from PIL import Image
            source_image = np.array(Image.open(source_image).convert("RGB").resize((width, height)))
        elif torch.is_tensor(source_image):
            source_image = source_image.detach().cpu().numpy()
        if not isinstance(source_image, np.ndarray) or source_image.ndim != 3 or source_image.shape[-1] != 3:
            raise ValueError("source_image must be a file path or an (H, W, 3) uint8 RGB array (EMOAnimationPipeline.py:686-689)")
        if source_image.shape[0] % 8 or source_image.shape[1] % 8:
            raise ValueError(f"source_image {source_image.shape[:2]} must be a multiple of 8 in both directions")
        return self.images2latents(source_image[None], None)

    # ------------------------------------------------------------------ the rest of the reference class's surface
    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator, latents=None, clip_length=16):
        """EMOAnimationPipeline.py:341-368: noise for `clip_length` frames, TILED video_length // clip_length times along the frame axis (so a
        video_length < 16 yields an empty tensor upstream - pass `latents` / `init_latents` there), times `scheduler.init_noise_sigma`.  Host
        RNG, like the reference (`torch.randn(shape, generator=generator)`)."""
        shape = (batch_size, num_channels_latents, clip_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        device = torch.device(device)
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn(shape, generator=generator[i], device=generator[i].device, dtype=dtype) for i in range(batch_size)], dim=0).to(device)
            else:
                latents = torch.randn(shape, generator=generator, device=generator.device if generator is not None else device, dtype=dtype).to(device)
            latents = latents.repeat(1, 1, video_length // clip_length, 1, 1)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def prepare_condition(self, condition, num_videos_per_prompt, device, dtype, do_classifier_free_guidance):
        """:370-377: uint8 (f, h, w, c) conditioning frames -> / 255, one copy per video, 'b f h w c -> (b f) c h w', doubled under CFG."""
        import numpy as np
        condition = torch.from_numpy(np.ascontiguousarray(condition).copy()).to(device=device, dtype=dtype) / 255.0
        condition = torch.stack([condition for _ in range(num_videos_per_prompt)], dim=0)
        condition = condition.permute(0, 1, 4, 2, 3).reshape(-1, condition.shape[4], condition.shape[2], condition.shape[3]).clone()
        return torch.cat([condition] * 2) if do_classifier_free_guidance else condition

    def select_controlnet_res_samples(self, controlnet_res_samples_cache_dict, context, do_classifier_free_guidance, b, f):
        """:514-540: the cached per-frame ControlNet residuals of the windows in `context`, '(b f) c h w -> b c f h w', repeated for CFG
        (the loop itself keeps this cache in rows - `_controlnet_residuals`; this is the reference's tensor-level helper)."""
        frames = [i for c in context for i in c]
        n_down = len(controlnet_res_samples_cache_dict[frames[-1]][0])
        down = [torch.cat([controlnet_res_samples_cache_dict[i][0][k] for i in frames]) for k in range(n_down)]
        mid = torch.cat([controlnet_res_samples_cache_dict[i][1] for i in frames])
        b = b // 2 if do_classifier_free_guidance else b

        def to5(t):
            t = t.reshape(b, f, *t.shape[1:]).permute(0, 2, 1, 3, 4)
            return t.repeat(2, 1, 1, 1, 1) if do_classifier_free_guidance else t
        return [to5(t) for t in down], to5(mid)

    @torch.no_grad()
    def decode_latents(self, latents, rank=0, decoder_consistency=None):
        """:291-307: latents / 0.18215 -> the VAE decoder frame by frame (or `decoder_consistency(frame_latents)`) -> (b, 3, f, H, W) in
        [0, 1] as a float32 numpy array."""
        if decoder_consistency is not None:
            f = latents.shape[2]
            lat = (1 / 0.18215 * latents).permute(0, 2, 1, 3, 4).reshape(-1, *latents.shape[1:2], *latents.shape[3:])
            video = torch.cat([decoder_consistency(lat[i:i + 1]) for i in range(lat.shape[0])])
            video = video.reshape(-1, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
            return (video / 2 + 0.5).clamp(0, 1).cpu().float().numpy()
        if self.vae is None:
            raise ValueError("decode_latents needs a VAE (emote_hack_amd.vae.AutoencoderKL)")
        return self.vae.decode_video(latents).cpu().float().numpy()

    def next_step(self, model_output, timestep: int, x, eta=0., verbose=False):
        """:379-400 - one step of DDIM INVERSION (x_t -> x_{t + T/n}): pred_x0 = (x - sqrt(1 - a) eps) / sqrt(a) with a taken one stride BELOW
        `timestep` (final_alpha_cumprod below 0), x_next = sqrt(a_next) pred_x0 + sqrt(1 - a_next) eps.  Returns (x_next, pred_x0).  Scalars
        on the host in f64; the two linear combinations run as emo_add launches on the device (plain torch for CPU tensors)."""
        sch = self.scheduler
        if verbose:
            print("timestep: ", timestep)
        nxt = int(timestep)
        t = min(nxt - sch.config.num_train_timesteps // sch.num_inference_steps, 999)
        a_t = float(sch.alphas_cumprod[t]) if t >= 0 else float(sch.final_alpha_cumprod)
        a_next = float(sch.alphas_cumprod[nxt])
        k0x, k0e = 1.0 / a_t ** 0.5, -((1.0 - a_t) ** 0.5) / a_t ** 0.5               # pred_x0 = k0x x + k0e eps
        knx, kne = a_next ** 0.5 * k0x, a_next ** 0.5 * k0e + (1.0 - a_next) ** 0.5    # x_next  = knx x + kne eps

        def lincomb(cx, ce):
            if not x.is_cuda:
                return cx * x + ce * model_output
            xf, ef = x.float().reshape(1, -1).contiguous(), model_output.float().reshape(1, -1).contiguous()
            z = ops.add(xf, ef, alpha=ce / cx)                     # x + (ce / cx) eps
            return ops.add(z, z, alpha=cx - 1.0).reshape(x.shape).to(x.dtype)      # ... times cx
        return lincomb(knx, kne), lincomb(k0x, k0e)

    @torch.no_grad()
    def invert(self, image, prompt=None, num_inference_steps=20, num_actual_inference_steps=10, eta=0.0, return_intermediates=False, **kwargs):
        """:417-477 - deterministic DDIM inversion of real frames into a noise map: frames -> latents (`images2latents`, or pass
        latents=(f, 4, h, w)), then for the ascending timesteps x <- next_step(unet(x as one (1, c, f, h, w) clip, t, text), t, x), stopping after
        `num_actual_inference_steps`.  The CLIP text encoder is outside this build: pass text_embeddings=(1, L, D) (or a text_encoder +
        tokenizer pair on the pipeline, called like upstream)."""
        text_embeddings = kwargs.get("text_embeddings")
        if text_embeddings is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("invert: pass text_embeddings=(1, L, D) (no CLIP text encoder in this build)")
            text_input = self.tokenizer(prompt, padding="max_length", max_length=77, return_tensors="pt")
            text_embeddings = self.text_encoder(text_input.input_ids.to(self._execution_device))[0]
        latents = kwargs.get("latents")
        if latents is None:
            latents = self.images2latents(image)
        latents = latents.to(self._execution_device)
        self.scheduler.set_timesteps(num_inference_steps)
        latents_list, pred_x0_list = [latents], [latents]
        for i, t in enumerate(reversed(self.scheduler.timesteps)):
            if num_actual_inference_steps is not None and i >= num_actual_inference_steps:
                continue
            model_inputs = latents.permute(1, 0, 2, 3).unsqueeze(0)                          # 'f c h w -> 1 c f h w'
            noise_pred = self.unet(model_inputs, int(t), encoder_hidden_states=text_embeddings).sample
            noise_pred = noise_pred[0].permute(1, 0, 2, 3)                                   # 'b c f h w -> (b f) c h w'
            latents, pred_x0 = self.next_step(noise_pred.to(latents.dtype), int(t), latents)
            latents_list.append(latents)
            pred_x0_list.append(pred_x0)
        if return_intermediates:
            return latents, latents_list
        return latents

    def interpolate_latents(self, latents: torch.Tensor, interpolation_factor: int, device):
        """:479-512: `interpolation_factor - 1` frames interpolated between every two consecutive frames with the method set by
        `set_tensor_interpolation_method` (magicanimate/utils/util.py:116-140: slerp over the whole frame tensor, linear when the two are
        nearly parallel).  `__call__` hard-codes the factor 1 (:824), for which this is the identity - outside the loop, host-level torch math."""
        if interpolation_factor < 2:
            return latents
        method = get_tensor_interpolation_method()
        if method is None:
            raise TypeError("'NoneType' object is not callable: call set_tensor_interpolation_method(is_slerp) first (magicanimate/utils/util.py:116-123)")
        n = latents.shape[2]
        out = torch.zeros((latents.shape[0], latents.shape[1], (n - 1) * interpolation_factor + 1, latents.shape[3], latents.shape[4]),
                          device=latents.device, dtype=latents.dtype)
        rate = [i / interpolation_factor for i in range(interpolation_factor)][1:]
        k = 0
        for i0 in range(n - 1):
            v0, v1 = latents[:, :, i0], latents[:, :, i0 + 1]
            out[:, :, k] = v0
            k += 1
            for fr in rate:
                out[:, :, k] = method(v0.to(device=device), v1.to(device=device), fr).to(latents.device)
                k += 1
        out[:, :, k] = latents[:, :, n - 1]
        return out

    # ------------------------------------------------------------------ reference-compatible entry point
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt=None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents=None, output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable] = None, callback_steps: Optional[int] = 1, controlnet_condition: list = None,
                 controlnet_conditioning_scale: float = 1.0, context_frames: int = 16, context_stride: int = 1,
                 context_overlap: int = 4, context_batch_size: int = 1, context_schedule: str = "uniform",
                 init_latents=None, num_actual_inference_steps: Optional[int] = None, appearance_encoder=None,
                 reference_control_writer=None, reference_control_reader=None, source_image=None,
                 decoder_consistency=None, audio=None, head_rotation_speeds=None, **kwargs):
        """Signature = EMOAnimationPipeline.py:544-578.  Extra keyword inputs for the parts that are out of
        scope here: text_embeddings=(2,L,D), ref_image_latents=(1,4,h,w), audio_features=(F,L_a,D),
        speed_embeddings=(1,4*C0), seed=int; dist/rank/world_size as in the reference (:636-638).  Execution knobs (all
        optional): use_graphs (default: HIP-graph replay on a HIP device - the path bench.py measures), reference_group
        (ReferenceNet timesteps per batched pass; default 10, the configuration bench.py measures - 25 would make two passes per 50-step clip,
        40.03 vs 40.38 ms per step), reference_lookahead, fusion_blocks, motion_latents, reuse_state
        (default True: a second call with the same geometry reuses the prepared plan and its captured graphs)."""
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        assert num_videos_per_prompt == 1   # :641
        if isinstance(prompt, list) and len(prompt) != 1:
            raise AssertionError("batch_size == 1")  # :642
        cn_cond = None
        if self.controlnet is not None:
            if controlnet_condition is None:
                raise ValueError("controlnet_condition (F,H,W,3 uint8 frames or a (F,3,H,W) float tensor) is required with a ControlNet")
            # prepare_condition (:369-376): uint8 HWC frames / 255 -> (f, c, h, w)
            cn_cond = controlnet_condition if torch.is_tensor(controlnet_condition) else torch.as_tensor(controlnet_condition)
            if cn_cond.dim() == 4 and cn_cond.shape[-1] == 3 and cn_cond.shape[1] != 3:
                cn_cond = cn_cond.permute(0, 3, 1, 2).float() / 255.0
            cn_cond = cn_cond.float()
        elif controlnet_condition is not None:
            raise ValueError("controlnet_condition given but the pipeline has no controlnet")
        if appearance_encoder is None:
            raise ValueError("appearance_encoder (ReferenceNet) is required")
        text_embeddings = kwargs.get("text_embeddings")
        if text_embeddings is None:
            if self.text_encoder is None:
                raise ValueError("pass text_embeddings=(2,L,D) [uncond, cond] (no CLIP text encoder in this build)")
            text_embeddings = self.text_encoder(prompt, negative_prompt)
        ref_lat = kwargs.get("ref_image_latents")
        if ref_lat is None:
            if self.vae is None or source_image is None:
                raise ValueError("pass ref_image_latents=(1,4,h,w) (no VAE in this build)")
            ref_lat = self._source_image_latents(source_image, width, height)   # :686-689 -> images2latents (:402-414)
        if audio is not None and kwargs.get("audio_features") is None:
            # :592-593 `audio_features = feature_extractor.extract_features_from_mp4(audio, m=2, n=2)` (a module-level extractor
            # there): here `audio` is the mono 16 kHz waveform (the container demux / resampling of Net.py:670-735 stays with the
            # caller) and feature_extractor= an emote_hack_amd.wav2vec2.Wav2VecFeatureExtractor with caller-loaded weights
            fx = kwargs.get("feature_extractor")
            if fx is None:
                raise ValueError("audio= needs feature_extractor= (emote_hack_amd.wav2vec2.Wav2VecFeatureExtractor), or pass audio_features=")
            if isinstance(audio, (str, bytes)):
                raise ValueError("audio= takes the decoded mono 16 kHz waveform (file reading / demuxing is outside this path)")
            from .conditioning import audio_context_tokens
            windows = fx.extract_features(audio, m=2, n=2)
            kwargs["audio_features"] = audio_context_tokens(windows, video_length, fx.model.config.hidden_size)
        if head_rotation_speeds is not None and kwargs.get("speed_embeddings") is None:
            raise NotImplementedError("pass speed_embeddings= (see emote_hack_amd.conditioning.SpeedEncoder)")
        if init_latents is not None:   # (b f) c h w -> b c f h w  (:657-658)
            bf, c4, hh, ww = init_latents.shape
            lat = init_latents.reshape(bf // video_length, video_length, c4, hh, ww).permute(0, 2, 1, 3, 4)
        elif latents is not None:
            lat = latents
        else:
            # prepare_latents draws clip_length=16 frames and tiles them video_length//16 times (:341-360)
            if video_length < 16:
                raise ValueError("video_length < 16 needs init_latents (prepare_latents tiles 16-frame noise, :341-360)")
            g = generator if isinstance(generator, torch.Generator) else None
            base = torch.randn(1, self.unet.in_channels, 16, height // 8, width // 8, generator=g)
            lat = base.repeat(1, 1, video_length // 16, 1, 1) * self.scheduler.init_noise_sigma
        lat = self.denoise(lat, ref_lat, text_embeddings, appearance_encoder=appearance_encoder,
                           num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, eta=eta,
                           context_frames=context_frames, context_stride=context_stride, context_overlap=context_overlap,
                           context_batch_size=context_batch_size, context_schedule=context_schedule,
                           audio_features=kwargs.get("audio_features"), speed_embeddings=kwargs.get("speed_embeddings"),
                           seed=kwargs.get("seed", 0), dist=kwargs.get("dist", False), rank=kwargs.get("rank", 0),
                           world_size=kwargs.get("world_size", 1), num_actual_inference_steps=num_actual_inference_steps,
                           callback=callback, callback_steps=callback_steps, controlnet=self.controlnet, controlnet_cond=cn_cond,
                           controlnet_conditioning_scale=controlnet_conditioning_scale,
                           # the measured path: HIP-graph replay (default on a HIP device), batched ReferenceNet groups, and the
                           # prepared state kept for the next clip of the same geometry
                           use_graphs=kwargs.get("use_graphs"), reference_group=kwargs.get("reference_group", 10),   # = the configuration bench.py measures (25: two passes per 50-step clip, -0.35 ms per step)
                           reference_lookahead=kwargs.get("reference_lookahead"), fusion_blocks=kwargs.get("fusion_blocks", "midup"),
                           motion_latents=kwargs.get("motion_latents"), reuse_state=kwargs.get("reuse_state", True),
                           text_pairing=kwargs.get("text_pairing", "reference"))
        if self.vae is not None and output_type != "latent":
            video = self.vae.decode_video(lat)   # caller-supplied (:291-307)
        else:
            video = lat
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)
