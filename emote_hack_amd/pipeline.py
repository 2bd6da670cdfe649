"""EMOAnimationPipeline - the sampling loop of the reference (EMOAnimationPipeline.py:543-840) on
MI355X.  Same `__call__` signature; the hot loop (`:698-823`) is re-designed:

  * per step: ReferenceNet write pass -> per window-batch Backbone read pass -> window average + CFG +
    scheduler step fused in ONE HIP kernel (emo_cfg_step) on f32 master latents;
  * ReferenceNet banks depend on the timestep only (never on the latents), so with world_size>1 the
    `num_inference_steps` write passes are dealt round-robin over the ranks BEFORE the loop and exchanged
    with one RCCL all_gather (north_star: "all-gather over xGMI to broadcast ReferenceNet features");
    288 GB HBM holds every step's banks (50 x 28 MB at 512^2).  world_size==1 keeps the reference's
    per-step order;
  * windows are sharded `global_context[rank::world_size]` exactly like the reference (:757); the
    reference's gather-to-root + broadcast (:796-821) is replaced by one all_reduce of the f32 window
    accumulators, after which every rank runs the (deterministic, counter-based-noise) sampler step
    redundantly - no broadcast, identical latents on all ranks.

Out of scope here: CLIP text encoder, VAE, ControlNet, wav2vec.  They are
accepted as caller-supplied callables / precomputed tensors (`text_embeddings=`, `ref_image_latents=`,
`audio_features=`, `speed_embeddings=` keyword arguments).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, List, Optional, Union

import os

import torch

from . import ops
from .context import get_context_scheduler
from .reference_control import ReferenceAttentionControl
from .scheduler import DDIMScheduler, DDPMScheduler


@dataclass
class AnimationPipelineOutput:  # EMOAnimationPipeline.py:79-81
    videos: Union[torch.Tensor, "object"]


class EMOAnimationPipeline:
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, controlnet=None, scheduler=None):
        """EMOAnimationPipeline.py:87-130.  The ctor forces steps_offset=1 / clip_sample=False on the
        scheduler it is given, like the reference."""
        if unet is None or scheduler is None:
            raise ValueError("unet and scheduler are required")
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.controlnet, self.scheduler = unet, controlnet, scheduler
        if isinstance(scheduler, DDIMScheduler) and scheduler.config.steps_offset != 1:
            scheduler.config.steps_offset = 1
        if getattr(scheduler.config, "clip_sample", False):
            scheduler.config.clip_sample = False
        self.vae_scale_factor = 8
        self.device = unet.device

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

    # ------------------------------------------------------------------ ReferenceNet banks
    def _write_banks(self, appearance_encoder, writer, ref_lat_rep, t, text_embeddings):
        """ReferenceNet write pass (:711-716).  The reference runs it on [uncond-text, cond-text] copies of the image, but
        the reader never uses the uncond bank row: for the uc rows `hidden_states_c` is overwritten by the bank-free
        uc attention (mutual_self_attention.py:243-256), and the ReferenceNet is batch-independent.  Only the cond half
        is computed; outputs are bit-identical (tests/test_gpu_unet.py loop parity against the full oracle)."""
        writer.clear()
        h = ref_lat_rep.shape[0] // 2
        appearance_encoder(ref_lat_rep[h:].unsqueeze(2), t, encoder_hidden_states=text_embeddings[h:], return_dict=False)
        return writer

    def _pack_banks(self, writer):
        """All bank tensors of one step in one contiguous buffer (one collective instead of ten)."""
        flat = [writer.bank[p][0].reshape(-1) for p in writer.order]
        return torch.cat(flat)  # device copy (plumbing)

    def _unpack_banks(self, buf, writer, shapes):
        off = 0
        for p, shp in zip(writer.order, shapes):
            n = shp[0] * shp[1] * shp[2]
            writer.bank[p] = [buf[off:off + n].view(shp)]
            off += n

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def prepare_denoise(self, latents, ref_image_latents, text_embeddings, *, appearance_encoder, num_inference_steps=50,
                        guidance_scale=7.5, eta=0.0, context_frames=16, context_stride=1, context_overlap=4,
                        context_batch_size=1, context_schedule="uniform", audio_features=None, speed_embeddings=None, seed=0,
                        fusion_blocks="midup", dist=False, rank=0, world_size=1, return_eps=False, use_graphs=False,
                        controlnet=None, controlnet_cond=None, controlnet_conditioning_scale=1.0):
        """Set up the loop state (EMOAnimationPipeline.py:628-696).  latents f32 (1,4,F_tot,h,w);
        ref_image_latents (1,4,h,w); text_embeddings (2,L,D) = [uncond, cond]."""
        unet, sch = self.unet, self.scheduler
        dev = unet.device
        if not guidance_scale > 1.0:
            raise NotImplementedError("guidance_scale <= 1 (no CFG) is not on the benchmarked path")
        if latents.shape[0] != 1:
            raise ValueError("batch_size must be 1 (EMOAnimationPipeline.py:641-642); run clips as separate calls")
        st = SimpleNamespace()
        st.cbs = context_batch_size
        st.latents = latents.to(dev).float().contiguous()
        _, st.C4, st.f_tot, st.h, st.w = st.latents.shape
        st.HW = st.h * st.w
        st.text = torch.cat([text_embeddings] * st.cbs).to(dev)                               # :631
        st.appearance_encoder = appearance_encoder
        st.writer = ReferenceAttentionControl(appearance_encoder, do_classifier_free_guidance=True, mode="write",
                                              batch_size=st.cbs, fusion_blocks=fusion_blocks)  # :633
        st.reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", batch_size=st.cbs,
                                              fusion_blocks=fusion_blocks)                      # :634
        st.num_inference_steps = num_inference_steps
        st.timesteps = sch.set_timesteps(num_inference_steps)
        st.ref_rep = ref_image_latents.to(dev).float().repeat(st.cbs * 2, 1, 1, 1)           # :712
        scheduler_fn = get_context_scheduler(context_schedule)
        # the reference recomputes the (step-independent: step arg is always 0) window list every step (:748-755)
        queue = list(scheduler_fn(0, num_inference_steps, st.f_tot, context_frames, context_stride, context_overlap))
        nb = math.ceil(len(queue) / st.cbs)
        st.global_context = [queue[i * st.cbs:(i + 1) * st.cbs] for i in range(nb)]
        st.frame_idx = {tuple(c): torch.tensor(c, dtype=torch.int32, device=dev) for ctx in st.global_context for c in ctx}
        # EMO_FORCE_DIST=1: take the multi-GPU code path even at world_size 1 (single-GPU functional test of that path)
        force_dist = bool(dist) and os.environ.get("EMO_FORCE_DIST") == "1"
        st.dist_pre, st.rank_pre, st.world_pre = (bool(dist) and world_size > 1) or force_dist, rank, world_size
        st.my_contexts = st.global_context[rank::world_size] if st.dist_pre else st.global_context          # :757
        st.ctx_index = [[torch.tensor(c, dtype=torch.int64, device=dev) for c in ctx] for ctx in st.my_contexts]
        st.t_table = torch.tensor(st.timesteps, dtype=torch.int64, device=dev)   # INT timestep table, bit-exact
        st.t_buf = torch.zeros(1, dtype=torch.int64, device=dev)
        st.use_graphs, st.graphs, st.graph_pool = bool(use_graphs), {}, None
        st.overlap, st.side, st.writer_pool, st.unet_state = False, None, None, {}
        if st.use_graphs:
            st.graph_pool = torch.cuda.graph_pool_handle()
            st.writer_pool = torch.cuda.graph_pool_handle()   # own pool: the write pass may run concurrently with the down path
            st.overlap = (not st.dist_pre) and fusion_blocks == "midup" and os.environ.get("EMO_NO_OVERLAP") != "1"
            st.side = torch.cuda.Stream() if st.overlap else None
        if audio_features is not None:
            audio_features = audio_features.to(dev)
        st.noise_pred = torch.empty(2, st.C4, st.f_tot, st.HW, device=dev, dtype=torch.float32)
        st.counter = torch.empty(st.f_tot, device=dev, dtype=torch.float32)
        st.guidance_scale, st.eta, st.seed = guidance_scale, eta, seed
        st.audio_features, st.speed_embeddings = audio_features, speed_embeddings
        st.dist, st.rank, st.world_size = (bool(dist) and world_size > 1) or force_dist, rank, world_size
        st.t_ref = torch.zeros(1, dtype=torch.int64, device=dev)   # timestep of the ReferenceNet pass this rank computes
        st.return_eps, st.eps_trace = return_eps, []
        st.bank_group, st.bank_shapes, st.bank_group_start, st.bank_now = None, None, -1, None
        # ControlNet branch (EMOAnimationPipeline.py:643-650,678-679,718-746): (F_tot,3,H,W) conditioning images in [0,1]
        st.controlnet = controlnet
        if controlnet is not None:
            if controlnet_cond is None or controlnet_cond.shape[0] != st.f_tot:
                raise ValueError("controlnet_cond must hold one (3,H,W) conditioning image per frame")
            st.cn_scale = float(controlnet_conditioning_scale)
            st.cn_cond = controlnet_cond.to(dev).float()
            # the frames this rank's windows touch; residuals are computed once per frame and step, in chunks of
            # context_frames (the reference caches them per frame from overlap-0 windows: the network is per-frame, so
            # the chunking does not enter the result)
            need = sorted({k for ctx in st.my_contexts for c in ctx for k in c})
            st.cn_frames = need
            st.cn_pos = {k: i for i, k in enumerate(need)}
            st.cn_chunks = [torch.tensor(need[i:i + context_frames], dtype=torch.int64, device=dev) for i in range(0, len(need), context_frames)]
            st.cn_text = st.text[st.text.shape[0] // 2:][:1]        # cond text embedding (:678-679)
            st.cn_sel = [[torch.tensor([st.cn_pos[k] for c in ctx for k in c], dtype=torch.int64, device=dev)] for ctx in st.my_contexts]
            st.cn_down, st.cn_mid = None, None
        return st

    def _exchange_banks(self, st, si):
        """Multi-GPU ReferenceNet hand-off.  The write pass depends on the timestep only, so the ranks deal
        the passes of the next `world_size` steps among themselves (rank r computes step si+r) and swap the
        packed banks with ONE all_gather over xGMI - each rank runs the ReferenceNet once per world_size
        steps instead of every step (the reference recomputes it on every rank, :711-716)."""
        import torch.distributed as td
        ws = st.world_size
        mine = min(si + st.rank, len(st.timesteps) - 1)   # tail group: surplus ranks recompute the last step (unused)
        st.t_ref.copy_(st.t_table[mine:mine + 1], non_blocking=True)
        # the pass goes through the same warm -> capture -> replay sequence as the UNet parts (its ~450 launches cost
        # ~20 ms of host time when enqueued from Python)
        self._run(st, "writer_dist", lambda: self._part_writer_dist(st), pool=st.writer_pool)
        send = st.send
        recv = torch.empty(ws * send.numel(), device=send.device, dtype=send.dtype)   # flat: valid for RCCL and gloo
        td.all_gather_into_tensor(recv, send)
        st.bank_group, st.bank_group_start = recv.view(ws, send.numel()), si
        if getattr(st, "bank_now", None) is None:
            st.bank_now = torch.empty_like(send)

    def _part_writer_dist(self, st):
        self._write_banks(st.appearance_encoder, st.writer, st.ref_rep, st.t_ref, st.text)
        st.bank_shapes = [tuple(st.writer.bank[p][0].shape) for p in st.writer.order]
        st.send = self._pack_banks(st.writer)

    # ---- the two GPU-heavy parts of a step; both read the timestep from the device buffer st.t_buf so that they
    #      can be captured once into HIP graphs and replayed for the other 49 steps
    def _part_writer(self, st):
        self._write_banks(st.appearance_encoder, st.writer, st.ref_rep, st.t_buf, st.text)    # :711-716

    def _part_controlnet(self, st):
        """Per-frame ControlNet residuals of this step (:718-746)."""
        downs, mids = [], []
        for idx in st.cn_chunks:
            x = self.scheduler.scale_model_input(st.latents.index_select(2, idx), None)[0].permute(1, 0, 2, 3).contiguous()
            d, m = st.controlnet(x, st.t_buf, encoder_hidden_states=st.cn_text.repeat(idx.numel(), 1, 1),
                                 controlnet_cond=st.cn_cond.index_select(0, idx), conditioning_scale=st.cn_scale, return_dict=False)
            downs.append(d)
            mids.append(m)
        st.cn_down = [torch.cat([d[i] for d in downs]) for i in range(len(downs[0]))]
        st.cn_mid = torch.cat(mids)

    def _controlnet_residuals(self, st, ci):
        """select_controlnet_res_samples (:514-540): frames of the window batch, '(b f) c h w -> b c f h w', CFG repeat."""
        if st.controlnet is None:
            return {}
        sel = st.cn_sel[ci][0]
        nb, nf = len(st.my_contexts[ci]), len(st.my_contexts[ci][0])

        def to5(t):
            y = t.index_select(0, sel)
            return y.reshape(nb, nf, *y.shape[1:]).permute(0, 2, 1, 3, 4).repeat(2, 1, 1, 1, 1)
        return dict(down_block_additional_residuals=tuple(to5(t) for t in st.cn_down), mid_block_additional_residual=to5(st.cn_mid))

    def _unet_inputs(self, st, ci):
        dev = self.unet.device
        idx = st.ctx_index[ci]                                                               # device int64 frame indices
        x = torch.cat([st.latents.index_select(2, i) for i in idx]).repeat(2, 1, 1, 1, 1)    # :759-763 (index/copy only)
        x = self.scheduler.scale_model_input(x, None)
        af = None
        if st.audio_features is not None:   # per-frame audio context; uc rows get a zero context
            cond = torch.cat([st.audio_features.index_select(0, i) for i in idx])
            af = torch.cat([torch.zeros_like(cond), cond])
        return x, af

    def _accumulate(self, st, ci, rows):
        context = st.my_contexts[ci]
        nf = len(context[0])
        for j, c in enumerate(context):                                                       # :790-794
            fr = st.frame_idx[tuple(c)]
            for branch in (0, 1):
                bi = branch * len(context) + j
                ops.accumulate_window(rows[bi * nf * st.HW:(bi + 1) * nf * st.HW], st.noise_pred[branch], st.counter, fr,
                                      C_=st.C4, F=st.f_tot, HW=st.HW, add_counter=(branch == 0))

    def _update_reader(self, st):
        if st.writer.order and not all(st.writer.bank[p] for p in st.writer.order):
            raise RuntimeError("ReferenceNet banks are empty at reader.update(): the write pass must run (or be captured) "
                               "in the same step as its consumer")
        st.reader.update(st.writer)                                                           # :774

    def _part_unet(self, st, ci):
        x, af = self._unet_inputs(st, ci)
        self._update_reader(st)
        rows = self.unet(x, st.t_buf, encoder_hidden_states=st.text[:x.shape[0]], audio_features=af,
                         speed_embeddings=st.speed_embeddings, return_dict=False, _return_rows=True,
                         **self._controlnet_residuals(st, ci))                                # :777-786
        st.reader.clear()                                                                     # :788
        self._accumulate(st, ci, rows)

    # the same pass split at the first use of the reference banks (fusion 'midup': mid block): the down path does not
    # depend on the ReferenceNet, so it overlaps the write pass running on a second HIP stream
    def _part_unet_down(self, st, ci):
        x, af = self._unet_inputs(st, ci)
        s = self.unet._begin(x, st.t_buf, st.text[:x.shape[0]], af, st.speed_embeddings, **self._controlnet_residuals(st, ci))
        self.unet._run_down(s)
        st.unet_state[ci] = s

    def _part_unet_rest(self, st, ci):
        s = st.unet_state[ci]
        self._update_reader(st)
        st.reader._prepare(s.c, self.unet)
        rows = self.unet._run_rest(s, return_rows=True)
        st.reader.clear()                                                                     # :788
        self._accumulate(st, ci, rows)

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
        st.t_buf.copy_(st.t_table[si:si + 1], non_blocking=True)     # device-to-device: the INT timestep of this step
        st.noise_pred.zero_()
        st.counter.zero_()
        if st.controlnet is not None:   # per-frame residual cache of this step (:718-746), before any UNet part needs it
            self._run(st, "controlnet", lambda: self._part_controlnet(st))
        overlap = st.overlap
        if overlap:
            # ReferenceNet write pass  ||  bank-independent Backbone down path.  All three parts go through the same
            # warm -> capture -> replay sequence IN THE SAME STEPS, so the reader/writer bank lists are populated when the
            # consumers are captured (a graph replay does not re-run the Python bank bookkeeping).  The second stream is
            # only used once graphs exist (eager allocations must not cross streams).
            use_side = st.graphs.get("writer") is not None and ops.PROFILER is None
            main = torch.cuda.current_stream()
            if use_side:
                st.side.wait_stream(main)
                with torch.cuda.stream(st.side):
                    self._run(st, "writer", lambda: self._part_writer(st), pool=st.writer_pool)
            else:
                self._run(st, "writer", lambda: self._part_writer(st), pool=st.writer_pool)
            for ci in range(len(st.my_contexts)):
                self._run(st, ("down", ci), lambda ci=ci: self._part_unet_down(st, ci))
            if use_side:
                main.wait_stream(st.side)
            for ci in range(len(st.my_contexts)):
                self._run(st, ("rest", ci), lambda ci=ci: self._part_unet_rest(st, ci))
        elif not st.dist:
            self._run(st, "writer", lambda: self._part_writer(st), pool=st.writer_pool)
        else:
            if st.bank_group is None or si >= st.bank_group_start + st.world_size or si < st.bank_group_start:
                self._exchange_banks(st, si)
            st.bank_now.copy_(st.bank_group[si - st.bank_group_start])   # static buffer: graph-stable addresses
            self._unpack_banks(st.bank_now, st.writer, st.bank_shapes)
        if not overlap:
            for ci in range(len(st.my_contexts)):                                              # :757
                self._run(st, ("unet", ci), lambda ci=ci: self._part_unet(st, ci))
        if st.dist:                                                                            # replaces :796-809 + :819-821
            import torch.distributed as td
            td.all_reduce(st.noise_pred)
            td.all_reduce(st.counter)
        c_x, c_eps, c_n = sch.coefficients(t, st.eta) if isinstance(sch, DDIMScheduler) else sch.coefficients(t)
        eps_out = torch.empty(st.C4 * st.f_tot * st.HW, device=dev, dtype=torch.float32) if st.return_eps else None
        ops.cfg_step(st.noise_pred, st.counter, st.latents, C_=st.C4, F=st.f_tot, HW=st.HW, guidance_scale=st.guidance_scale,
                     c_x=c_x, c_eps=c_eps, c_noise=c_n, seed=st.seed, step=si, eps_out=eps_out)  # :812-817 fused
        if st.return_eps:
            st.eps_trace.append(eps_out.view(1, st.C4, st.f_tot, st.h, st.w))
        st.writer.clear()                                                                      # :823

    @torch.no_grad()
    def denoise(self, latents, ref_image_latents, text_embeddings, *, num_actual_inference_steps=None, callback=None,
                callback_steps=1, **kw):
        """The whole loop; returns the denoised latents f32 (1,4,F_tot,h,w) (and the eps trace if asked)."""
        st = self.prepare_denoise(latents, ref_image_latents, text_embeddings, **kw)
        n = st.num_inference_steps
        for si, t in enumerate(st.timesteps):
            if num_actual_inference_steps is not None and si < n - num_actual_inference_steps:   # :699-700
                continue
            self.denoise_step(st, si)
            if callback is not None and si % callback_steps == 0:
                callback(si, t, st.latents)
        return (st.latents, st.eps_trace) if st.return_eps else st.latents

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
        speed_embeddings=(1,4*C0), seed=int; dist/rank/world_size as in the reference (:636-638)."""
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
            ref_lat = self.vae.encode(source_image).latent_dist.mean * 0.18215   # :402-414
        if audio is not None and kwargs.get("audio_features") is None:
            raise NotImplementedError("wav2vec feature extraction is out of scope: pass audio_features=")
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
                           controlnet_conditioning_scale=controlnet_conditioning_scale)
        if self.vae is not None and output_type != "latent":
            video = self.vae.decode_video(lat)   # caller-supplied (:291-307)
        else:
            video = lat
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)
