#!/usr/bin/env python3
"""bench.py - denoised frames/s of the Emote-hack diffusion hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--mode weak|strong]
        N>1: one rank per GPU over RCCL - either launched by torch.distributed.run (RANK / WORLD_SIZE in the environment), or,
        when WORLD_SIZE is not set, bench.py re-executes ITSELF under `python -m torch.distributed.run --nnodes=1
        --nproc-per-node N --master-addr 127.0.0.1` (the reference spawns one process per GPU the same way,
        magicanimate/pipelines/animation.py:246-269); rank 0 prints the one JSON line.
        --mode strong = BASELINE configs[3]: ONE 48-frame clip (4 windows x 2 CFG branches = 8 units) whatever N is -
        one unit per rank at N = 8, two at N = 4, all eight on one GPU at N = 1.

Workload (BASELINE.json configs[1], "cfg2"): 512x512 -> latent 64x64, 12-frame window, 50-step DDPM,
classifier-free guidance 7.5 (uncond + cond branch batched), ReferenceNet on (midup banks), AnimateDiff motion
modules at every resolution, text context (1,77,768), bf16, synthetic latents + name-keyed random weights of
the SD-1.5 architecture (1277 M-parameter Backbone + 860 M-parameter ReferenceNet).

A "step" = ONE iteration of the sampling loop over one batch of synthetic input: Backbone UNet forward on the rank's
(window x CFG-branch) units, eps hand-off, window accumulate, fused CFG + scheduler step - everything
EMOAnimationPipeline.py:698-823 does per timestep - plus its share of the ReferenceNet work: the write pass is batched
over REF_GROUP = 10 timesteps (the banks depend on the timestep only) and launched once per 10 steps on a second stream,
one group ahead of the Backbone.  Any window of K consecutive steps contains K/10 such passes (the driver's --steps 20
contains two), i.e. exactly the per-step share of a full 50-step run; the final synchronise covers both streams.
Inputs are resident in HBM when the timed region starts.  value = frames / (num_inference_steps * s_per_step), whole job.
Weak scaling: an N-GPU run denoises a 12*N-frame clip as N windows = 2N (window x branch) units; rank r owns both
branches of window r (U[r::N] of the branch-major unit list), one all_gather of the eps slices per step, a group's
ReferenceNet timesteps dealt over the ranks + one all_gather of the banks per group.

Adds to the JSON line:  "roofline" for the dominant kernel (live HIP-event timing of every launch of that
kernel family in an eager pass right after the timed region; algorithmic FLOPs per launch = 2*M*N*K for the GEMM/conv
kernel) and "cpu_baseline" (the CPU oracle timed on the host cores on a bounded sample, rank 0, N=1 only); "parity": the full-size oracle
forward that leg computes, used as the CHECKER of the HIP output in the benchmarked dtype and in f32 mode (max / mean abs error, north_star's
rtol 1e-3 / atol 1e-4 verdict); config.clocks: SCLK / MCLK of the device during the timed region (amdgpu sysfs);
config.whole_clips_after_timed_region: --clips whole 50-step clips run back to back behind the timed region (not `value`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
NUM_INFERENCE_STEPS = 50
REF_GROUP = 10
# algorithmic TFLOP per UNet forward at cfg2 (cond with ReferenceNet K/V, uncond, ReferenceNet image)
TFLOP_COND, TFLOP_UNCOND, TFLOP_REFNET = 14.319, 13.251, 0.803
# BASELINE.json configs (SURVEY.md 8d for the FLOP figures): latent size, frames per window, compute dtype, per-frame audio context,
# speed-layer embeddings, algorithmic TFLOP per (cond, uncond, ReferenceNet image) forward
CONFIGS = {
    "cfg2": dict(hw=64, f_win=12, dtype="bf16", audio=False, speed=False, tflop=(14.319, 13.251, 0.803),
                 name="cfg2 (BASELINE configs[1]): 512x512 latents 64x64, 12-frame window per GPU"),
    # (the 5-token audio context replaces the 77-token text context of the cond units' attn2: ~0.07 TFLOP less per forward; priced at the cfg2 figure)
    "cfg3": dict(hw=64, f_win=12, dtype="bf16", audio=True, speed=False, tflop=(14.319, 13.251, 0.803),
                 name="cfg3 (BASELINE configs[2]): cfg2 + per-frame wav2vec audio tokens (5 x 768) as the attn2 context of the cond units (train_stage_2 path)"),
    "cfg5": dict(hw=96, f_win=24, dtype="f16", audio=True, speed=True, tflop=(77.394, 67.625, 2.148),
                 name="cfg5 (BASELINE configs[4]): 768x768 latents 96x96, 24-frame window per GPU, fp16, speed-layer embedding + per-frame audio "
                      "context (train_stage_3 path)"),
}


def build_models(dev, dtype):
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.spec import param_shapes
    from emote_hack_amd.synth import synth_state_dict
    from emote_hack_amd.unet import UNet3DConditionModel
    from tests import cases
    unet = UNet3DConditionModel(**cases.SD15_MOTION)
    unet.load_state_dict(synth_state_dict(param_shapes(unet.spec), device=dev))
    unet.to(dev, dtype)
    ref = AppearanceEncoderModel(**cases.SD15)
    ref.load_state_dict(synth_state_dict(param_shapes(ref.spec), prefix=cases.REF_PREFIX, device=dev))
    ref.to(dev, dtype)
    return unet, ref


def cpu_baseline(unet, ref, full=False):
    """The oracle (kind 'port': plain-PyTorch fp32 CPU restatement, pinned by the reference's goldens) on the host cores,
    bounded sample (~25 s): WARM timings (second of two runs) of one uncond Backbone forward on a 2-frame 512^2 window and
    of one ReferenceNet forward, extrapolated like the reference would run cfg2: per step one uncond + one cond forward of
    the 12-frame window (conv / linear / spatial attention scale linearly in F - 6x the 2-frame time; cond carries the
    reference K/V: x 14.319 / 13.251 by FLOPs) + the ReferenceNet on two copies of the image (EMOAnimationPipeline.py:711-716),
    once per window, not per frame.  Also one cold 1-frame forward on 8 threads (the survey's thread count) and BASELINE
    configs[0] ("cfg1": 256x256 single frame, no motion module) in full.  Returns (parity, baseline): with full=True the 12-frame
    forward's OUTPUT is handed to parity_vs_oracle as the reference of the HIP path."""
    from oracle import unet_ref as U
    from tests import cases
    from emote_hack_amd.synth import seeded_randn
    cores = min(os.cpu_count() or 1, 32)   # threads actually used (oversubscribing a 256-thread host is 30x slower)
    torch.set_num_threads(cores)
    sd_u = {k: v.float().cpu() for k, v in unet.state_dict().items()}
    sd_r = {k: v.float().cpu() for k, v in ref.state_dict().items()}
    Fs = 2
    x, ctx = seeded_randn((1, 4, Fs, 64, 64), 1), seeded_randn((1, 77, 768), 2)

    def timed(fn, warm=True):
        if warm:
            fn()
        t0 = time.time()
        fn()
        return time.time() - t0
    with torch.no_grad():
        t_unet = timed(lambda: U.unet_forward(sd_u, cases.SD15_MOTION, x, 981, ctx))
        t_ref = timed(lambda: U.unet_forward(sd_r, cases.SD15, x[:, :, :1], 981, ctx, bank_mode="write"))
        sd_1 = {k: v for k, v in sd_u.items() if "motion_modules" not in k}
        x1 = seeded_randn((1, 4, 1, 32, 32), 1)
        t_cfg1 = timed(lambda: U.unet_forward(sd_1, cases.SD15, x1, 981, ctx))
        torch.set_num_threads(min(8, cores))
        t_unet8 = timed(lambda: U.unet_forward(sd_u, cases.SD15_MOTION, x[:, :, :1], 981, ctx), warm=False)
        torch.set_num_threads(cores)
    F_WIN = 12
    t_uncond = t_unet * F_WIN / Fs
    t_full = None
    parity = None
    if full:   # ONE whole 12-frame cfg2 uncond forward (temporal attention ~ F^2, the 5-D GroupNorm working set)
        xf = seeded_randn((1, 4, F_WIN, 64, 64), 1)
        kept = []
        with torch.no_grad():
            t_full = timed(lambda: kept.append(U.unet_forward(sd_u, cases.SD15_MOTION, xf, 981, ctx)), warm=False)
        t_uncond = t_full
        try:
            parity = parity_vs_oracle(unet, xf, ctx, kept[0])
        except Exception as ex:   # the checker must never cost the baseline figure
            parity = {"error": f"{type(ex).__name__}: {ex}"}
    t_cond = t_uncond * TFLOP_COND / TFLOP_UNCOND
    t_step = t_uncond + t_cond + 2 * t_ref
    t_step8 = (t_unet8 * F_WIN) * (1 + TFLOP_COND / TFLOP_UNCOND) + 2 * t_ref * (t_unet8 / (t_unet / Fs))
    return parity, {"value": F_WIN / (NUM_INFERENCE_STEPS * t_step), "unit": "denoised frames/s", "cores": cores, "kind": "port",
            "sample": (f"oracle fp32: ONE full 12-frame 512x512 uncond UNet fwd ({t_full:.1f}s, cold) + 1 ReferenceNet fwd ({t_ref:.1f}s, warm); step = uncond + "
                       f"1.08 x uncond (cond, by FLOPs) + 2 x refnet = {t_step:.0f}s, x{NUM_INFERENCE_STEPS} steps" if full else
                       f"oracle fp32, warm: 1 uncond UNet fwd on a {Fs}-frame 512x512 window ({t_unet:.1f}s) + 1 ReferenceNet fwd "
                       f"({t_ref:.1f}s); 12-frame step = 6 x uncond + 6.48 x (cond) + 2 x refnet = {t_step:.0f}s, x{NUM_INFERENCE_STEPS} steps"),
            "value_8_threads": F_WIN / (NUM_INFERENCE_STEPS * t_step8),
            "sample_8_threads": f"1 cold 1-frame uncond fwd on 8 threads ({t_unet8:.1f}s), same extrapolation",
            "full_12_frame_uncond_forward_s": t_full,
            "cfg1_seconds_per_forward": t_cfg1, "cfg1": "BASELINE configs[0]: (1,4,1,32,32), t=981, ctx 77x768, no motion module, full forward"}


def parity_vs_oracle(unet, x, ctx, y_ref):
    """The forward the cpu_baseline leg has just paid for, used as the CHECKER: the same 12-frame 512x512 uncond UNet evaluation
    (t = 981) on the HIP path - in the benchmarked dtype and in the f32 validation mode of the same kernels - against the oracle's
    output.  north_star's tolerance (rtol 1e-3 / atol 1e-4) is a statement about the f32 mode; the low-precision figure is reported
    next to it, not hidden in the smoke log."""
    from emote_hack_amd.unet import UNet3DConditionModel
    from tests import cases
    dev = unet.device
    out = {"what": "one full-size cfg2 uncond UNet forward (1,4,12,64,64), t=981, ctx (1,77,768): HIP vs the CPU oracle's fp32 output",
           "ref_mean_abs": float(y_ref.abs().mean()), "ref_max_abs": float(y_ref.abs().max())}

    def one(m):
        y = m(x.to(dev), 981, ctx.to(dev)).sample.float().cpu()
        e = (y - y_ref).abs()
        return {"max_abs": float(e.max()), "mean_abs": float(e.mean()),
                "within_rtol1e-3_atol1e-4": bool(torch.allclose(y, y_ref, rtol=1e-3, atol=1e-4))}
    name = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}[unet.dtype]
    out["dtype"] = name
    out.update(one(unet))
    if unet.dtype != torch.float32:
        m32 = UNet3DConditionModel(**cases.SD15_MOTION)
        m32.load_state_dict(unet._master)
        m32.to(dev, torch.float32)
        out["f32"] = one(m32)
        del m32
        torch.cuda.empty_cache()
    return out


def _sysfs_card(index=0):
    """/sys/class/drm/cardN/device of the torch device `index`: matched by PCI address (a box of this pool shows all 8 cards of its
    node in sysfs while the container sees one GPU); None if it cannot be told."""
    import glob
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        return None
    try:
        pr = torch.cuda.get_device_properties(index)
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        for c in cards:
            if want in os.path.realpath(os.path.dirname(c)).lower():
                return os.path.dirname(c)
    except Exception:
        pass
    return os.path.dirname(cards[0]) if len(cards) == 1 else None


_CARD = {}


def gpu_clocks(index=0):
    """Current shader / memory clock of the GPU (MHz) from the amdgpu sysfs tables (the line marked '*' in pp_dpm_sclk /
    pp_dpm_mclk); None where the node is not readable.  Compute-bound kernels move with SCLK, HBM-bound ones do not - boxes of
    this pool differ by ~8 % on the same build, so the line says which clocks it was measured at."""
    import re
    if index not in _CARD:
        _CARD[index] = _sysfs_card(index)
    base = _CARD[index]
    if base is None:
        return None
    out = {"card": os.path.basename(os.path.dirname(base))}
    for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        try:
            cur = [ln for ln in open(os.path.join(base, fn)).read().splitlines() if ln.strip().endswith("*")]
            out[key] = int(re.search(r"(\d+)\s*mhz", cur[0], re.I).group(1)) if cur else None
        except Exception:
            out[key] = None
    return out


class ClockSampler:
    """Samples gpu_clocks() every 50 ms on a host thread while a region runs (sysfs reads, no GPU work)."""

    def __init__(self, index=0):
        import threading
        self.index, self.samples, self._stop = index, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            c = gpu_clocks(self.index)
            if c:
                self.samples.append(c)
            self._stop.wait(0.05)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join()

    def summary(self):
        def agg(k):
            v = [s[k] for s in self.samples if s.get(k)]
            return {"min": min(v), "max": max(v), "mean": sum(v) / len(v)} if v else None
        return {"samples": len(self.samples), "card": self.samples[0].get("card") if self.samples else None,
                "sclk_mhz": agg("sclk_mhz"), "mclk_mhz": agg("mclk_mhz")}


def _profile_of_this_build(pattern):
    """The committed counter profile (profiles/<pattern>) taken on THE RUNNING BUILD: tools/pmc_to_json.py stamps every
    profile with the sha256 of the kernel sources (emote_hack_amd.build.csrc_digest); a profile of another build (or an
    unstamped one from an earlier round) is NOT attached - a stale counter figure next to a fresh timing misleads."""
    import glob
    from emote_hack_amd.build import csrc_digest
    want = csrc_digest()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("csrc_sha256") == want:
            return f, d
    return None, None


def pmc_traffic(family):
    """HBM bytes per launch of the kernel family from the committed rocprofv3 PMC passes of THIS build (FETCH_SIZE / WRITE_SIZE
    collected in separate runs of this same command, FETCH x2 gfx950 correction) - profiles/r*_pmc.json.  PMC counters
    cannot be read from inside the process: the figure is tagged with the profile it came from; null when no committed
    profile matches the running kernel sources."""
    f, d = _profile_of_this_build("r*_pmc.json")
    if d is None:
        return None
    e = d.get("per_kernel_family", {}).get(family)
    return {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "source": os.path.relpath(f, ROOT), "csrc_sha256": d["csrc_sha256"][:16],
            "note": "from the committed profile named in `source` (same command, same kernel sources, earlier run), not from this run"} if e else None


def mfma_util_from_profiles():
    """MFMA utilisation of the path from the committed SQ counter pass of THIS build (profiles/r*_mfma.json,
    tools/pmc_to_json.py): SQ_VALU_MFMA_BUSY_CYCLES over the dispatch cycles of all 1024 SIMDs - tagged with its source; null
    when no committed profile matches the running kernel sources."""
    f, d = _profile_of_this_build("r*_mfma.json")
    if d is None:
        return None
    return {"mfma_busy_frac": d["mfma_busy_frac"], "source": os.path.relpath(f, ROOT), "csrc_sha256": d["csrc_sha256"][:16]}


def _rccl_version():
    try:
        return ".".join(map(str, torch.cuda.nccl.version()))
    except Exception as ex:   # (evidence only: never cost the measurement)
        return f"unavailable ({type(ex).__name__})"


def spawn_command(argv, n, port=None):
    """The command bench.py re-executes itself as when --gpus N > 1 and no launcher set WORLD_SIZE: one rank per GPU under
    torch.distributed.run on 127.0.0.1 (the container's hostname may not resolve)."""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_check(rank, world):
    """`--spawn-check`: prove the launch geometry without a GPU - every rank joins a gloo group and rank 0 prints the count."""
    import torch.distributed as td
    td.init_process_group("gloo")
    t = torch.ones(1)
    td.all_reduce(t)
    if rank == 0:
        print(json.dumps({"spawned_ranks": int(t.item()), "world_size": world}))
    td.barrier()
    td.destroy_process_group()


def bench_vae(a, dev, dtype):
    """--stage vae: decode_latents (EMOAnimationPipeline.py:291-307) of one denoised 12-frame 512x512 clip on the HIP AutoencoderKL
    (SD-1.x geometry, 83.7 M parameters, name-keyed random weights): latents / 0.18215 -> per-frame decoder -> (x / 2 + 0.5).clamp
    -> (1, 3, 12, 512, 512) f32.  A "step" = one whole decode_video call; value = decoded frames/s."""
    from emote_hack_amd import ops
    from emote_hack_amd.synth import seeded_randn, synth_state_dict
    from emote_hack_amd.vae import AutoencoderKL, VAE_DEFAULTS, vae_param_shapes
    vae = AutoencoderKL()
    vae.load_state_dict(synth_state_dict(vae_param_shapes(dict(VAE_DEFAULTS)), prefix="vae.", device=dev))
    vae.to(dev, dtype)
    F = 12
    lat = (0.18215 * seeded_randn((1, 4, F, 64, 64), 9)).to(dev)
    for _ in range(max(1, a.warmup if a.warmup < 3 else 2)):
        out = vae.decode_video(lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = max(1, min(a.steps, 10))
    for _ in range(steps):
        out = vae.decode_video(lat)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / steps
    res = {"metric": "VAE-decoded frames/s (512x512, decode_latents)", "value": F / dt_s, "unit": "frames/s", "n_gpus": 1, "steps": steps,
           "warmup": 2, "ms_per_step": dt_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
           "data": "synthetic",
           "config": {"workload": "SURVEY 8(f) row 2: AutoencoderKL.decode_video of a (1,4,12,64,64) latent clip -> (1,3,12,512,512), "
                                  "4 frames per decoder call, eager launches", "ms_per_frame": dt_s * 1e3 / F,
                      "output_finite": bool(torch.isfinite(out).all())}}
    if not a.no_profile:
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        vae.decode_video(lat)
        summ = prof.summary()
        ops.PROFILER = None
        dom = max((k for k in summ if summ[k]["flops"] > 0), key=lambda k: summ[k]["ms"])
        d = summ[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / MFMA_PEAK_BF16_TFLOPS, "traffic": None, "launches_per_step": d["launches"],
                           "avg_launch_us": d["ms"] * 1e3 / d["launches"]}
        res["kernels"] = {k: {"launches_per_step": v["launches"], "ms_per_step": v["ms"],
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None,
                              "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in sorted(summ.items())}
        res["config"]["executed_tflop_per_frame"] = sum(v["flops"] for v in summ.values()) / 1e12 / F
        if os.environ.get("EMO_BENCH_SHAPES"):
            rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1]["ms"])
            with open(os.environ["EMO_BENCH_SHAPES"], "w") as f:
                f.write("| kernel | shape | launches/clip | ms/clip | TFLOP/s | GB/s (algorithmic) |\n|---|---|---|---|---|---|\n")
                for (name, tag), v in rows[:40]:
                    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
                    f.write(f"| {name} | {tag} | {v['launches']:.1f} | {v['ms']:.3f} | {tf:.0f} | {v['bytes'] / (v['ms'] * 1e-3) / 1e9:.0f} |\n")
    print(json.dumps(res))


def bench_call(a, pipe, ref, dev, rank, world, dist, F_WIN, f_tot):
    """--entry call: the drop-in entry point (EMOAnimationPipeline.py:544-578, called by magicanimate/pipelines/animation.py:197-214)
    timed END TO END for whole clips - everything `pipe(...)` does up to the denoised latents: plan / graph capture on the first
    call, input binding + context K/V + 50 loop iterations + the ReferenceNet groups on every call."""
    from emote_hack_amd.synth import seeded_randn
    if dist:
        import torch.distributed as td
    kw = dict(video_length=f_tot, height=512, width=512, num_inference_steps=NUM_INFERENCE_STEPS, guidance_scale=7.5,
              context_frames=F_WIN, context_stride=1, context_overlap=0, output_type="latent", appearance_encoder=ref,
              text_embeddings=seeded_randn((2, 77, 768), 2), ref_image_latents=seeded_randn((1, 4, 64, 64), 3), seed=0,
              dist=dist, rank=rank, world_size=world, reference_group=a.ref_group, use_graphs=False if a.no_graphs else None)
    lat0 = seeded_randn((1, 4, f_tot, 64, 64), 1).to(dev)

    def sync():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
    times = []
    for i in range(1 + max(1, a.calls)):
        sync()
        t0 = time.perf_counter()
        out = pipe("", latents=lat0, **kw).videos
        sync()
        times.append(time.perf_counter() - t0)
    if dist:
        t = torch.tensor(times, device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        times = [float(x) for x in t.tolist()]
    cold, warm = times[0], times[1:]
    dt_s = sum(warm) / len(warm)
    if rank == 0:
        print(json.dumps({
            "metric": "denoised frames/s (512x512, 50-step DDPM)", "value": f_tot / dt_s, "unit": "frames/s", "n_gpus": world,
            "steps": NUM_INFERENCE_STEPS * len(warm), "warmup": NUM_INFERENCE_STEPS, "ms_per_step": dt_s / NUM_INFERENCE_STEPS * 1e3,
            "higher_is_better": True, "scaling": a.mode, "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "cfg2 through EMOAnimationPipeline.__call__ (output_type='latent'): whole 50-step clips, 512x512 latents "
                                   "64x64, 12-frame window per GPU, CFG 7.5, ReferenceNet on; the first call (plan + HIP-graph capture) is "
                                   "the warm-up, the timed calls reuse the prepared plan",
                       "entry": "EMOAnimationPipeline.__call__", "frames_total": f_tot, "num_inference_steps": NUM_INFERENCE_STEPS,
                       "hip_graphs": not a.no_graphs, "cold_call_s": cold, "warm_call_s": warm,
                       "cold_call_ms_per_step": cold / NUM_INFERENCE_STEPS * 1e3, "latents_finite": bool(torch.isfinite(out).all())}}))
    if dist:
        td.barrier()
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32", "f16"], help="compute dtype (default: the config's - bf16, cfg5 fp16)")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json configuration: cfg2 (the headline metric), cfg3 (+ audio "
                    "context), cfg5 (768x768 x 24 frames, fp16, speed + audio)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event instrumentation pass")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel from Python instead of replaying HIP graphs")
    ap.add_argument("--ref-group", type=int, default=REF_GROUP, help="ReferenceNet timesteps per batched pass")
    ap.add_argument("--no-ln-fold", action="store_true", help="A/B: explicit LayerNorm launches instead of the GEMM fold")
    ap.add_argument("--no-shared-prefix", action="store_true", help="A/B: compute the shared prefix of the [uncond, cond] batch for both halves")
    ap.add_argument("--no-fused-tail", action="store_true", help="A/B: ff.net.2 and proj_out as two GEMMs instead of one over [g | h]")
    ap.add_argument("--gn-fold-min-hw", type=int, default=None, help="A/B: pixels per frame from which the transformer GroupNorm is folded into proj_in (0 = never)")
    ap.add_argument("--gn-two-launch", action="store_true", help="A/B: every GroupNorm as statistics + apply launches (no one-launch kernel)")
    ap.add_argument("--no-merge-qkv", action="store_true", help="A/B: attn1's q | k and V^T projections as two launches instead of one")
    ap.add_argument("--gn-conv-min-hw", type=int, default=None, help="A/B: pixels per frame from which the resnets' GroupNorm + SiLU run inside the 3x3 conv (0 = never)")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"],
                    help="weak: a 12-frame window per GPU (12*N frames); strong: BASELINE configs[3], one 48-frame clip = 8 units over N GPUs")
    ap.add_argument("--spawn-check", action="store_true", help="only prove that N ranks start (gloo, no GPU needed)")
    ap.add_argument("--entry", default="step", choices=["step", "call"],
                    help="step: time K loop iterations of a prepared state (the contract's default); call: time the reference-compatible "
                         "entry point EMOAnimationPipeline.__call__ end to end for WHOLE 50-step clips (output_type='latent'): one "
                         "cold call (plan + graph capture), then --calls timed calls that reuse the prepared plan")
    ap.add_argument("--calls", type=int, default=2, help="--entry call: timed calls after the cold one")
    ap.add_argument("--clips", type=int, default=None, help="whole 50-step clips run back to back BEHIND the timed region and reported as "
                    "config.whole_clips_after_timed_region (default 3; cfg5: 1; 0 = skip)")
    ap.add_argument("--stage", default="loop", choices=["loop", "vae"],
                    help="loop: the sampling loop (the headline metric); vae: the step right behind it - decode_latents of the 12-frame "
                         "512x512 clip (EMOAnimationPipeline.py:291-307) on the HIP AutoencoderKL, ms per frame (SURVEY 8f row 2)")
    ap.add_argument("--controlnet", action="store_true",
                    help="the loop with the ControlNet branch on (EMOAnimationPipeline.py:718-746, SURVEY 8f row 1): SD-1.5-sized "
                         "ControlNetModel on 512x512 conditioning images, residuals cached per frame and step")
    ap.add_argument("--cpu-baseline-sample", action="store_true",
                    help="cpu_baseline from a 2-frame forward scaled x6 (~25 s of CPU time) instead of the default: ONE full 12-frame cfg2 uncond "
                         "forward of the oracle (BASELINE.md section 4; ~60 s in all)")
    ap.add_argument("--emulate-rank", default=None, metavar="R/N",
                    help="MEASUREMENT aid on one GPU: time the work of rank R of an N-rank job (its units, its share of every ReferenceNet group, the "
                         "projection of the whole group) without collectives - a rank's critical path per step, i.e. what N GPUs would give if the "
                         "exchanges were free.  e.g. --mode strong --emulate-rank 4/8 = one cond unit of BASELINE configs[3].  NOT a multi-GPU number")
    ap.add_argument("--share-gpu", action="store_true", help="validation on a 1-GPU box: every rank on cuda:0, exchange through gloo "
                                                             "(RCCL takes one device per rank) - the number is NOT a multi-GPU measurement")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves (replaces this process; the ranks inherit stdout, rank 0 prints the JSON line)
        cmd = spawn_command(sys.argv[1:], a.gpus)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's hipIpc handles need it on this host driver
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
    if a.spawn_check:
        return spawn_check(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.share_gpu:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=dev)

    from emote_hack_amd import DDPMScheduler, ops
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    from emote_hack_amd.synth import seeded_randn

    cfg = CONFIGS[a.config]
    a.dtype = a.dtype or cfg["dtype"]
    if a.clips is None:
        a.clips = 1 if a.config == "cfg5" else 3
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[a.dtype]
    T_COND, T_UNCOND, T_REF = cfg["tflop"]
    if a.no_ln_fold:
        from emote_hack_amd import unet as unet_mod
        unet_mod.FOLD_LAYERNORM = False
    if a.no_shared_prefix:
        from emote_hack_amd import unet as unet_mod
        unet_mod.SHARE_CFG_PREFIX = False
    if a.no_fused_tail:
        from emote_hack_amd import unet as unet_mod
        unet_mod.FUSE_FF_TAIL = False
    if a.gn_fold_min_hw is not None:
        from emote_hack_amd import unet as unet_mod
        unet_mod.GN_FOLD_MIN_HW = a.gn_fold_min_hw
    if a.gn_two_launch:
        from emote_hack_amd import ops as ops_mod
        ops_mod.GN_ONE_LAUNCH = False
    if a.no_merge_qkv:
        from emote_hack_amd import unet as unet_mod
        unet_mod.MERGE_QKV = False
    if a.gn_conv_min_hw is not None:
        from emote_hack_amd import unet as unet_mod
        unet_mod.GN_CONV_MIN_HW = a.gn_conv_min_hw
    if a.stage == "vae":
        return bench_vae(a, dev, dtype)
    unet, ref = build_models(dev, dtype)
    F_WIN, HW_LAT = cfg["f_win"], cfg["hw"]
    emu = None
    if a.emulate_rank:
        emu = tuple(int(v) for v in a.emulate_rank.split("/"))
        if world != 1 or not 0 <= emu[0] < emu[1]:
            raise SystemExit("--emulate-rank R/N runs in ONE process (--gpus 1), 0 <= R < N")
    f_tot = F_WIN * (emu[1] if emu else world) if a.mode == "weak" else 4 * F_WIN       # strong: BASELINE configs[3] - 48 frames = 4 windows
    if a.config != "cfg2" and (a.entry != "step" or a.controlnet or a.mode != "weak"):
        raise SystemExit("--config cfg3 / cfg5 run the default loop measurement (--entry step, --mode weak, no ControlNet)")
    extra_kw = {}
    if cfg["audio"]:   # per-frame audio tokens (Net.py:649-667 windows of wav2vec features, projected to the context width)
        extra_kw["audio_features"] = seeded_randn((f_tot, 5, 768), 4)
    if cfg["speed"]:   # speed-bucket embedding added to the time embedding (train_stage_3_speedlayers.py:242-271)
        extra_kw["speed_embeddings"] = 0.1 * seeded_randn((1, 4 * 320), 5)
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDPMScheduler())
    if a.entry == "call":
        return bench_call(a, pipe, ref, dev, rank, world, dist, F_WIN, f_tot)
    cn_kw = {}
    if a.controlnet:   # SD-1.5-sized ControlNet (the Backbone's encoder geometry, magicanimate/models/controlnet.py:94-243)
        from emote_hack_amd import ControlNetModel
        from emote_hack_amd.spec import param_shapes
        from emote_hack_amd.synth import synth_state_dict
        from tests import cases
        cn = ControlNetModel(**dict(cases.SD15, down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)))
        cn.load_state_dict(synth_state_dict(param_shapes(cn.spec), prefix="controlnet.", device=dev))
        cn.to(dev, dtype)
        cn_kw = dict(controlnet=cn, controlnet_cond=seeded_randn((f_tot, 3, 512, 512), 77).clamp(-1, 1) * 0.5 + 0.5,
                     controlnet_conditioning_scale=1.0)
    st = pipe.prepare_denoise(seeded_randn((1, 4, f_tot, HW_LAT, HW_LAT), 1).to(dev), seeded_randn((1, 4, HW_LAT, HW_LAT), 3),
                              seeded_randn((2, 77, 768), 2), appearance_encoder=ref, num_inference_steps=NUM_INFERENCE_STEPS,
                              guidance_scale=7.5, context_frames=F_WIN, context_stride=1, context_overlap=0, seed=0,
                              dist=dist, rank=rank, world_size=world, use_graphs=not a.no_graphs, reference_group=a.ref_group,
                              _emulate_rank=emu, **cn_kw, **extra_kw)
    n_win = (emu[1] if emu else world) if a.mode == "weak" else 4
    if emu is None:
        assert len(st.windows) == n_win and len(st.units) == 2 * n_win
        assert sum(len(c.units) for c in st.calls) == (2 if a.mode == "weak" else len(st.units[rank::world]))

    def sync():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()

    si = 0
    # warm-up: W untimed steps (at least 2 with graphs: one eager pass, one capture pass)
    for _ in range(max(a.warmup, 0 if a.no_graphs else 2)):
        pipe.denoise_step(st, si % NUM_INFERENCE_STEPS)
        si += 1
    sync()
    clk_idle = gpu_clocks(local_rank)
    with ClockSampler(local_rank) as clk_timed:
        t0 = time.perf_counter()
        groups_before = st.groups_launched
        for _ in range(a.steps):
            pipe.denoise_step(st, si % NUM_INFERENCE_STEPS)
            si += 1
        host_s = time.perf_counter() - t0   # host time to ENQUEUE the steps (launches are asynchronous)
        sync()
        dt_s = time.perf_counter() - t0
    ref_passes = st.groups_launched - groups_before
    # WHOLE CLIPS right behind the timed region (not `value`): --clips x 50 loop iterations back to back, every ReferenceNet group
    # of a clip included - the per-step figure of a complete 50-step clip, long enough (seconds) for an external utilisation
    # sampler to see the GPU busy
    whole = None
    if a.clips > 0:
        while si % NUM_INFERENCE_STEPS:      # start the clips at step 0
            pipe.denoise_step(st, si % NUM_INFERENCE_STEPS)
            si += 1
        sync()
        with ClockSampler(local_rank) as clk_clip:
            tc = time.perf_counter()
            for _ in range(a.clips * NUM_INFERENCE_STEPS):
                pipe.denoise_step(st, si % NUM_INFERENCE_STEPS)
                si += 1
            sync()
            clip_s = time.perf_counter() - tc
        if dist:
            t = torch.tensor([clip_s], device=dev, dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            clip_s = float(t.item())
        whole = {"clips": a.clips, "steps": a.clips * NUM_INFERENCE_STEPS, "seconds": clip_s,
                 "ms_per_step": clip_s / (a.clips * NUM_INFERENCE_STEPS) * 1e3, "frames_per_s": a.clips * f_tot / clip_s,
                 "clocks": clk_clip.summary()}
    # per-kernel roofline pass: the SAME steps launched eagerly with every launch bracketed by HIP events on the launch
    # stream (a graph replay cannot host per-launch events); kernels and shapes are identical to the timed region.  The
    # ReferenceNet group pass (one per REF_GROUP steps) is profiled separately and enters per step with weight 1/REF_GROUP.
    prof = prof_ref = None
    if not a.no_profile:
        prof = ops.KernelProfiler()
        ops.PROFILER = prof
        prof_steps = min(a.steps, 2)
        base, glen = st.group_ready * st.T, len(st.groups[st.group_ready])   # steps of the resident group: no ReferenceNet work
        for i in range(prof_steps):
            pipe.denoise_step(st, base + min(1 + i, glen - 1))
        sync()
        prof_ref = ops.KernelProfiler()
        ops.PROFILER = prof_ref
        pipe._reference_group(st, st.group_ready)      # recompute the resident group in place (same values)
        sync()
        ops.PROFILER = None
    if dist:
        t = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt_s = float(t.item())
    finite = bool(torch.isfinite(st.latents).all())
    ms_per_step = dt_s / a.steps * 1e3
    fps = f_tot / (NUM_INFERENCE_STEPS * dt_s / a.steps)

    if rank == 0:
        out = {
            "metric": "denoised frames/s (512x512, 50-step DDPM)", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": a.mode,
            "vs_baseline": None, "dtype": a.dtype, "data": (f"synthetic; --emulate-rank {a.emulate_rank}: ONE rank's work on one GPU without collectives - `value` is the "
                                                            "N-GPU rate IF the exchanges were free and every rank as fast as this one (an upper bound, NOT a measurement)") if emu else
            "synthetic" if not a.share_gpu else "synthetic; --share-gpu: ranks time-share ONE GPU over gloo (logic validation, not a multi-GPU number)",
            "config": {"workload": (cfg["name"] + ", 50-step DDPM, CFG 7.5 (uc+c batched), "
                                    if a.mode == "weak" else
                                    "cfg4 (BASELINE configs[3]): 512x512 latents 64x64, ONE 48-frame clip = 4 windows of 12 x 2 CFG branches "
                                    f"= 8 units over {world} GPU(s), 50-step DDPM, CFG 7.5, ") +
                                   ("ControlNet branch ON (SD-1.5-sized, 512x512 conditioning images, per-frame residual cache), " if a.controlnet else "") +
                                   "ReferenceNet on (midup), motion modules res 1/2/4/8, ctx 77x768; 1 step = 1 loop iteration "
                                   f"+ 1/{st.T} of a {st.T}-timestep ReferenceNet pass",
                       "frames_total": f_tot, "num_inference_steps": NUM_INFERENCE_STEPS,
                       "parallelism": (f"(window x CFG-branch) units x{world}: rank r owns both branches of window r; " if a.mode == "weak" else
                                       f"8 (window x CFG-branch) units dealt U[r::{world}] ({len(st.units[rank::world])} per rank); ") +
                                      "all_gather of eps slices per step, all_gather of ReferenceNet banks per group",
                       "latents_finite": finite, "host_enqueue_ms_per_step": host_s / a.steps * 1e3,
                       "hip_graphs": not a.no_graphs, "reference_group": st.T,
                       "reference_group_note": "the same group size EMOAnimationPipeline.__call__ uses by default",
                       "reference_passes_in_timed_region": ref_passes,
                       "reference_passes_fair_share": a.steps / st.T,
                       "distributed": None if not dist else {
                           "backend": td.get_backend(), "world_size": td.get_world_size(),
                           "rccl_version": _rccl_version() if td.get_backend() == "nccl" else None,
                           "devices_visible": torch.cuda.device_count(), "one_device_per_rank": not a.share_gpu,
                           "collectives_per_step": "1 all_gather_into_tensor of the eps slices (main stream, default group)",
                           "collectives_per_reference_group": f"{len(st.bank_variants)} all_gather_into_tensor of the fp16-rounded banks (main stream, default group)"},
                       "clocks": {"idle_before": clk_idle, "timed_region": clk_timed.summary(),
                                  "source": "amdgpu sysfs pp_dpm_sclk / pp_dpm_mclk, sampled every 50 ms on a host thread"},
                       "whole_clips_after_timed_region": whole},
        }
        # algorithmic work per step: cond + uncond Backbone per window + the ReferenceNet pass.  The reference
        # runs the ReferenceNet on [uncond-text, cond-text] copies of the image (2 x 0.803 TFLOP); the uncond copy's features
        # are never read (mutual_self_attention.py:243-256 overwrites the uc rows) and everything behind the last bank write
        # is dead, so this path computes less.  The achieved rate is priced on the SURVEY figure for the cond copy only.
        tflop_ref = n_win * (T_COND + T_UNCOND) + 2 * T_REF
        tflop_step = n_win * (T_COND + T_UNCOND) + 1 * T_REF
        out["config"]["name"] = a.config
        out["config"]["algorithmic_tflop_per_frame_reference"] = NUM_INFERENCE_STEPS * tflop_ref / f_tot
        out["config"]["algorithmic_tflop_per_step_reference"] = tflop_ref
        out["config"]["algorithmic_tflop_per_step"] = tflop_step
        out["config"]["achieved_tflops_whole_path"] = None if emu else tflop_step / (dt_s / a.steps) / world
        out["config"]["whole_path_frac_of_mfma_peak"] = None if emu else tflop_step / (dt_s / a.steps) / world / MFMA_PEAK_BF16_TFLOPS
        if emu:
            out["config"]["emulated_rank"] = {"rank": emu[0], "world": emu[1], "units": [list(u) for c in st.calls for u in c.units],
                                              "reference_lookahead": st.lookahead, "reference_timesteps_per_group_this_rank": -(-st.T // emu[1])}
        if prof is not None:
            summ, summ_ref = prof.summary(), prof_ref.summary()
            merged = {}
            for src, wgt in ((summ, 1.0 / prof_steps), (summ_ref, 1.0 / st.T)):
                for k, v in src.items():
                    m = merged.setdefault(k, dict(launches=0.0, ms=0.0, flops=0.0, bytes=0.0))
                    for f in m:
                        m[f] += v[f] * wgt
            dom = max((k for k in merged if merged[k]["flops"] > 0), key=lambda k: merged[k]["ms"])
            d = merged[dom]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            out["roofline"] = {"kernel": {"gemm_dense": "gemm_kernel<bf16,false>", "gemm_conv3x3": "gemm_kernel<bf16,true>",
                                          "attention": "attention_kernel", "temporal_attention": "temporal_attention_mfma_kernel"}[dom],
                               "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / MFMA_PEAK_BF16_TFLOPS, "traffic": pmc_traffic(dom),
                               "launches_per_step": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                               "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                               "algorithmic_mbytes_per_launch": d["bytes"] / d["launches"] / 1e6,
                               "mfma_util": mfma_util_from_profiles()}
            out["roofline"]["measured_in"] = (f"eager HIP-event pass of {prof_steps} steps + one ReferenceNet group pass (weight 1/{st.T}) right "
                                              "after the timed region (same kernels/shapes)")
            out["kernels"] = {k: {"launches_per_step": v["launches"], "ms_per_step": v["ms"],
                                  "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None,
                                  "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in sorted(merged.items())}
            out["config"]["executed_tflop_per_step"] = sum(v["flops"] for v in merged.values()) / 1e12
            out["config"]["reference_group_kernel_ms"] = sum(v["ms"] for v in summ_ref.values())
        if prof is not None and os.environ.get("EMO_BENCH_SHAPES"):
            rows = {}
            for src, wgt in ((prof.by_shape(), 1.0 / prof_steps), (prof_ref.by_shape(), 1.0 / st.T)):
                for k, v in src.items():
                    m = rows.setdefault(k, dict(launches=0.0, ms=0.0, flops=0.0, bytes=0.0))
                    for f in m:
                        m[f] += v[f] * wgt
            rows = sorted(rows.items(), key=lambda kv: -kv[1]["ms"])
            with open(os.environ["EMO_BENCH_SHAPES"], "w") as f:
                f.write("| kernel | shape | launches/step | ms/step | TFLOP/s | GB/s (algorithmic) |\n|---|---|---|---|---|---|\n")
                for (name, tag), v in rows[:70]:
                    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
                    f.write(f"| {name} | {tag} | {v['launches']:.1f} | {v['ms']:.3f} | {tf:.0f} | "
                            f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:.0f} |\n")
        if world == 1 and not a.no_cpu_baseline and a.config == "cfg2":   # (the baseline leg times the cfg2 forward)
            try:
                out["parity"], out["cpu_baseline"] = cpu_baseline(unet, ref, full=not a.cpu_baseline_sample)
            except Exception as ex:   # the baseline leg must never cost the measurement
                out["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"}
        print(json.dumps(out))
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
