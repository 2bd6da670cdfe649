#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own in-tree code (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/gen_golden.py [--skip-cfg1]

Imports /root/reference through the plumbing-only diffusers shim (diffusers_shim.py), loads
name-keyed synthetic weights (emote_hack_amd/synth.py) into the reference modules, feeds seeded
inputs (tests/cases.py) and stores ONLY tensors / integers: no reference source, bytecode or
module travels to the GPU box.  The fixtures pin oracle/ (tests/test_oracle_golden.py) and are the
end-to-end targets of the HIP path (tests/test_gpu_*.py).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import diffusers_shim as shim  # noqa: E402

shim.install()

from magicanimate.models import attention as ref_attention  # noqa: E402
from magicanimate.models import embeddings as ref_emb  # noqa: E402
from magicanimate.models import motion_module as ref_mm  # noqa: E402
from magicanimate.models import orig_attention as ref_oa  # noqa: E402
from magicanimate.models import resnet as ref_resnet  # noqa: E402
from magicanimate.models.mutual_self_attention import ReferenceAttentionControl  # noqa: E402
from magicanimate.models.unet_controlnet import UNet3DConditionModel  # noqa: E402
from magicanimate.pipelines.context import uniform as ref_uniform  # noqa: E402

from emote_hack_amd.synth import seeded_randn, synth_state_dict  # noqa: E402
from tests import cases  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)
GOLD = cases.GOLDEN_DIR
os.makedirs(GOLD, exist_ok=True)


def load_synth(module, prefix=""):
    sd = module.state_dict()
    params = {k for k, _ in module.named_parameters()}
    new = synth_state_dict({k: tuple(v.shape) for k, v in sd.items() if k in params}, prefix=prefix)
    for k, v in sd.items():  # buffers (pos_encoder.pe, SpeedController.centers/radii) keep the module's own value
        if k not in params:
            new[k] = v
    module.load_state_dict(new, strict=True)
    return module.eval()


def key_listing(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def listing_digest(listing):
    canon = "\n".join(f"{k}:{','.join(map(str, listing[k]))}" for k in sorted(listing))
    return hashlib.sha256(canon.encode()).hexdigest()


def sorted_blocks(unet, fusion="midup"):
    """BasicTransformerBlocks in the reference's own pairing order (mutual_self_attention.py:532-537)."""
    from magicanimate.models.stable_diffusion_controlnet_reference import torch_dfs
    mods = torch_dfs(unet.mid_block) + torch_dfs(unet.up_blocks) if fusion == "midup" else torch_dfs(unet)
    mods = [m for m in mods if isinstance(m, ref_attention.BasicTransformerBlock)]
    return sorted(mods, key=lambda x: -x.norm1.normalized_shape[0])


def module_names(unet, mods):
    rev = {id(m): n for n, m in unet.named_modules()}
    return [rev[id(m)] for m in mods]


def gutted_listing():
    """The ReferenceNet ctor (magicanimate/models/appearance_encoder.py:217-633) cannot run here (its blocks are diffusers
    2-D blocks), but what it does to `up_blocks[3].attentions[2]` is plain in-tree Python (:613-621): read those assignments
    by AST - target path, replacement kind - and instantiate the two in-tree replacement classes to count their parameters.
    The product's AppearanceEncoderModel must own the F=1 UNet's keys minus everything under a replaced path."""
    import ast
    path = "/root/reference/magicanimate/models/appearance_encoder.py"
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AppearanceEncoderModel")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    repl = shim.extract_classes(path, ["Identity", "_LoRACompatibleLinear"], extra_ns={"LoRALinearLayer": object})

    def dotted(t):
        if isinstance(t, ast.Attribute):
            base = dotted(t.value)
            return t.attr if base == "self" else f"{base}.{t.attr}"
        if isinstance(t, ast.Subscript):
            return f"{dotted(t.value)}.{ast.literal_eval(t.slice)}"
        if isinstance(t, ast.Name):
            return t.id
        raise ValueError(ast.dump(t))

    def kind(v):
        if isinstance(v, ast.Constant) and v.value is None:
            return "None", 0
        if isinstance(v, ast.Call) and isinstance(v.func, ast.Name):
            m = repl[v.func.id]()
            return v.func.id, sum(p.numel() for p in m.parameters())
        if isinstance(v, ast.Call) and isinstance(v.func, ast.Attribute) and v.func.attr == "ModuleList":
            ks = [kind(e) for e in v.args[0].elts]
            return "ModuleList[" + ",".join(k for k, _ in ks) + "]", sum(n for _, n in ks)
        raise ValueError(ast.dump(v))

    rows = []
    for n in ast.walk(init):
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Attribute):
            try:
                tp = dotted(n.targets[0])
            except ValueError:
                continue
            if tp.startswith("up_blocks.") and ".attentions." in tp:
                k, npar = kind(n.value)
                rows.append([tp, k, int(npar)])
    assert rows, "no gutted-block assignments found"
    return rows


# =============================================================================== ints
def gen_ints():
    out = {}
    out["windows"] = [dict(args=list(c), windows=[list(map(int, w)) for w in
                      ref_uniform(0, 50, c[0], c[1], c[2], c[3])]) for c in cases.WINDOW_CASES]
    out["windows_step"] = [dict(step=s, args=[24, 16, 2, 4], windows=[list(map(int, w)) for w in
                           ref_uniform(s, 50, 24, 16, 2, 4)]) for s in (1, 2, 3, 7)]
    # SpeedController buckets (train_stage_3_speedlayers.py) via AST-extracted class
    S3 = shim.extract_classes("/root/reference/train_stage_3_speedlayers.py", ["SpeedController", "FaceRegionController"])
    sc = S3["SpeedController"](9, 16)
    out["speed_buckets"] = dict(speeds=cases.SPEEDS,
                                buckets=sc.map_speed_to_bucket(torch.tensor(cases.SPEEDS)).tolist())
    # state-dict key listings
    tiny = UNet3DConditionModel(**cases.TINY_MOTION)
    out["tiny_motion_keys"] = key_listing(tiny)
    tiny_lin = UNet3DConditionModel(**cases.TINY_LINEAR)
    out["tiny_linear_keys"] = key_listing(tiny_lin)
    with torch.device("meta"):
        full = UNet3DConditionModel(**cases.SD15_MOTION)
        full_mid = UNet3DConditionModel(**dict(cases.SD15_MOTION, motion_module_mid_block=True))
        full_plain = UNet3DConditionModel(**cases.SD15)
    for name, m in (("sd15_motion", full), ("sd15_motion_mid", full_mid), ("sd15", full_plain)):
        lst = key_listing(m)
        out[name + "_digest"] = dict(n_keys=len(lst), sha256=listing_digest(lst),
                                     n_params=int(sum(p.numel() for p in m.parameters())))
    out["appearance_encoder_gutted"] = gutted_listing()
    # A3 regression pin (NOT a reference pin: diffusers is absent, parity unpinned): timestep tables (INT) and the per-step
    # update coefficients of oracle/scheduler_ref.py, so that a silent change of either restatement is caught
    from oracle.scheduler_ref import SchedulerRef
    out["scheduler_tables"] = {}
    # "ddim" / "ddpm": steps_offset 1, as the pipeline ctor forces it on every scheduler that carries the key
    # (EMOAnimationPipeline.py:105-117); "ddpm0": a DDPM scheduler stepped outside the pipeline (diffusers' default offset 0)
    for name, kind, off in (("ddim", "ddim", 1), ("ddpm", "ddpm", 1), ("ddpm0", "ddpm", 0)):
        for n in (50, 25, 3):
            sch = SchedulerRef(kind, steps_offset=off)
            ts = sch.set_timesteps(n)
            out["scheduler_tables"][f"{name}_{n}"] = dict(timesteps=[int(t) for t in ts],
                                                          coefficients=[[float(c) for c in sch.coefficients(t)] for t in ts])
    # pairing order of the reference-attention banks
    out["bank_order_tiny_midup"] = module_names(tiny, sorted_blocks(tiny, "midup"))
    out["bank_order_tiny_full"] = module_names(tiny, sorted_blocks(tiny, "full"))
    out["bank_order_sd15_midup"] = module_names(full, sorted_blocks(full, "midup"))
    json.dump(out, open(os.path.join(GOLD, "ints.json"), "w"), indent=1, sort_keys=True)
    print("ints.json", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


# =============================================================================== modules
def gen_modules():
    T = {}
    # A5 timestep embedding
    ts = torch.tensor([981, 1, 500, 0, 999])
    T["temb/sinusoid320"] = ref_emb.Timesteps(320, True, 0)(ts)
    T["temb/sinusoid32_noflip_shift1"] = ref_emb.Timesteps(32, False, 1)(ts)
    te = load_synth(ref_emb.TimestepEmbedding(32, 128), "time_embedding.")
    T["temb/mlp_out"] = te(ref_emb.Timesteps(32, True, 0)(ts))
    # A7 resnet (with and without shortcut)
    for name, cin, cout in (("resnet_sc", 32, 64), ("resnet_id", 64, 64)):
        m = load_synth(ref_resnet.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=128, groups=8,
                                                eps=1e-5, non_linearity="silu"), name + ".")
        x, emb = seeded_randn((2, cin, 4, 8, 8), 11), seeded_randn((2, 128), 12)
        T[f"{name}/out"] = m(x, emb)
    # A8 samplers
    m = load_synth(ref_resnet.Downsample3D(64, use_conv=True, padding=1, name="op"), "down.")
    T["down/out"] = m(seeded_randn((1, 64, 3, 8, 8), 13))
    m = load_synth(ref_resnet.Upsample3D(64, use_conv=True), "up.")
    T["up/out"] = m(seeded_randn((1, 64, 3, 4, 4), 14))
    # A12 attention at the real head dims (8 heads x 40 / 80 / 160), self + cross
    for d in (40, 80, 160):
        c = 8 * d
        m = load_synth(ref_oa.CrossAttention(query_dim=c, heads=8, dim_head=d), f"attn{d}.")
        T[f"attn{d}/self_out"] = m(seeded_randn((1, 64, c), 20 + d))
        m = load_synth(ref_oa.CrossAttention(query_dim=c, cross_attention_dim=768, heads=8, dim_head=d), f"xattn{d}.")
        T[f"attn{d}/cross_out"] = m(seeded_randn((1, 64, c), 20 + d), seeded_randn((1, 7, 768), 21 + d))
    # A13 feed-forward
    m = load_synth(ref_oa.FeedForward(64, activation_fn="geglu"), "ff.")
    T["ff/out"] = m(seeded_randn((2, 16, 64), 30))
    # A10/A11 Transformer3DModel (conv and linear projection)
    for name, lin in (("tf3d", False), ("tf3d_lin", True)):
        m = load_synth(ref_attention.Transformer3DModel(4, 16, in_channels=64, num_layers=1, cross_attention_dim=32,
                                                        norm_num_groups=8, use_linear_projection=lin,
                                                        unet_use_cross_frame_attention=False,
                                                        unet_use_temporal_attention=False), name + ".")
        T[f"{name}/out"] = m(seeded_randn((2, 64, 3, 4, 4), 40), seeded_randn((2, 5, 32), 41)).sample
        # per-frame context (B*F rows) passes through un-repeated (attention.py:118-119)
        T[f"{name}/out_perframe_ctx"] = m(seeded_randn((2, 64, 3, 4, 4), 40), seeded_randn((6, 5, 32), 42)).sample
    # A16 motion module
    m = load_synth(ref_mm.VanillaTemporalModule(in_channels=64, **cases.MOTION_KW_TINY), "motion.")
    T["motion/out"] = m(seeded_randn((2, 64, 4, 4, 4), 50), None, None)
    m = load_synth(ref_mm.VanillaTemporalModule(in_channels=320, **cases.MOTION_KW_FULL), "motion320.")
    T["motion320/out"] = m(seeded_randn((1, 320, 12, 4, 4), 51), None, None)
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "modules.safetensors"))
    print("modules.safetensors", len(T), "tensors", sum(v.numel() for v in T.values()) * 4 / 1e6, "MB")


# =============================================================================== tiny UNet end to end
def run_writer(ref_unet, ref_lat, t, ctx, fusion="midup"):
    """ReferenceNet pass = the no-motion 3-D UNet at F=1 in 'write' mode."""
    ctl = ReferenceAttentionControl(ref_unet, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks=fusion)
    ref_unet(ref_lat.unsqueeze(2), t, ctx)
    blocks = sorted_blocks(ref_unet, fusion)
    banks = [b.bank[0].clone() for b in blocks]
    ctl.clear()
    for b in blocks:  # un-hook
        b.forward = b._original_inner_forward
    return banks


def run_reader(unet, x, t, ctx, banks, fusion="midup", cfg=True, batch_size=1, bank_dtype=None, **kw):
    """bank_dtype: the reference's update() hands fp16 banks to the reader whatever the model dtype (:588); a bf16 model then
    fails inside torch.cat / the to_k Linear (bf16 x fp16 promotes to f32) - the low-precision yard-stick of the bf16 mode
    therefore rounds through fp16 like update() and THEN casts to the model dtype (what the product's bank hand-off does)."""
    ctl = ReferenceAttentionControl(unet, do_classifier_free_guidance=cfg, mode="read", batch_size=batch_size, fusion_blocks=fusion)
    blocks = sorted_blocks(unet, fusion)
    for b, v in zip(blocks, banks):  # what update() does (mutual_self_attention.py:588)
        b.bank = [v.clone().to(torch.float16) if bank_dtype is None else v.clone().to(torch.float16).to(bank_dtype)]
    y = unet(x, t, ctx, **kw).sample
    ctl.clear()
    for b in blocks:
        b.forward = b._original_inner_forward
    return y


def gen_unet_tiny():
    T = {}
    x, ctx = cases.tiny_inputs(2, 4)
    # (a) no motion module, F=2
    u0 = load_synth(UNet3DConditionModel(**cases.TINY))
    T["plain/out"] = u0(x[:, :, :2], 981, ctx).sample
    # (b) motion modules on, F=4, tensor timestep
    u1 = load_synth(UNet3DConditionModel(**cases.TINY_MOTION))
    T["motion/out"] = u1(x, torch.tensor(961), ctx).sample
    # (b2) linear projection + upcast + per-level heads
    u2 = load_synth(UNet3DConditionModel(**cases.TINY_LINEAR))
    T["linear/out"] = u2(x, 500, ctx).sample
    # (b3) ControlNet-style additive residuals (unet_controlnet.py:430-447)
    res_shapes = [(2, 32, 4, 16, 16)] * 3 + [(2, 32, 4, 8, 8)] + [(2, 64, 4, 8, 8)] * 2 + [(2, 64, 4, 4, 4)] * 3 + \
                 [(2, 64, 4, 2, 2)] * 3
    down_res = tuple(0.1 * seeded_randn(s, 100 + i) for i, s in enumerate(res_shapes))
    mid_res = 0.1 * seeded_randn((2, 64, 4, 2, 2), 99)
    T["motion/out_ctrl"] = u1(x, 961, ctx, down_block_additional_residuals=down_res,
                              mid_block_additional_residual=mid_res).sample
    # (c) ReferenceNet write -> banks -> Backbone read with CFG batch 2
    ref = load_synth(UNet3DConditionModel(**cases.TINY), cases.REF_PREFIX)
    ref_lat = seeded_randn((1, 4, 16, 16), 3).repeat(2, 1, 1, 1)
    banks = run_writer(ref, ref_lat, 961, ctx)
    for i, b in enumerate(banks):
        T[f"banks/{i}"] = b
    T["read/out"] = run_reader(u1, x, 961, ctx, banks)
    # (c2) fusion_blocks="full" (mutual_self_attention.py:532-537): the down-path transformer blocks write / read banks too
    banks_full = run_writer(ref, ref_lat, 961, ctx, fusion="full")
    for i, b in enumerate(banks_full):
        T[f"banks_full/{i}"] = b
    T["read/out_full"] = run_reader(u1, x, 961, ctx, banks_full, fusion="full")
    # (b4) H / W not a multiple of 2^num_upsamplers: the upsamplers interpolate to the skip's size (unet_controlnet.py:357-365,456-459)
    T["motion/out_20x12"] = u1(seeded_randn((1, 4, 2, 20, 12), 1), 500, seeded_randn((1, 5, 32), 2)).sample
    # (d) the reference's OWN low-precision forwards of the same models / inputs: the yard-stick for the bf16 / fp16 HIP
    # modes (their error against the fp32 goldens is compared with the error of these tensors against the fp32 goldens)
    import copy
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        try:
            um = copy.deepcopy(u1).to(dt)
            T[f"motion/out_{name}"] = um(x.to(dt), torch.tensor(961), ctx.to(dt)).sample.float()
            up = copy.deepcopy(u0).to(dt)
            T[f"plain/out_{name}"] = up(x[:, :, :2].to(dt), 981, ctx.to(dt)).sample.float()
            T[f"read/out_{name}"] = run_reader(um, x.to(dt), 961, ctx.to(dt), [b.to(dt) for b in banks],
                                               bank_dtype=None if dt == torch.float16 else dt).float()
        except Exception as ex:   # a CPU op without a half kernel
            print(f"reference forward in {name} failed: {ex}")
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "unet_tiny.safetensors"))
    print("unet_tiny.safetensors", {k: tuple(v.shape) for k, v in T.items() if not k.startswith("banks")})
    return u1, ref


# =============================================================================== UNet forward switches off the shipped configs
def gen_unet_switches():
    """The two forward switches of unet_controlnet.py that no shipped config turns on and that need no third-party code:
    center_input_sample (:371-373) and the class embedding in its three ctor forms (:119-127,400-408) - the reference model itself on
    the tiny motion config."""
    T = {}
    x, ctx = cases.tiny_inputs(2, 4)
    u = load_synth(UNet3DConditionModel(**dict(cases.TINY_MOTION, center_input_sample=True)))
    T["center/out"] = u(x, 961, ctx).sample
    u = load_synth(UNet3DConditionModel(**dict(cases.TINY_MOTION, num_class_embeds=7)))
    T["class_table/out"] = u(x, 961, ctx, class_labels=torch.tensor([3, 5])).sample
    u = load_synth(UNet3DConditionModel(**dict(cases.TINY_MOTION, class_embed_type="timestep")))
    T["class_timestep/out"] = u(x, 961, ctx, class_labels=torch.tensor([10, 500])).sample
    u = load_synth(UNet3DConditionModel(**dict(cases.TINY_MOTION, class_embed_type="identity")))
    T["class_identity/out"] = u(x, 961, ctx, class_labels=0.1 * seeded_randn((2, 128), 71)).sample
    try:
        u(x, 961, ctx)
        raise SystemExit("the reference accepted a missing class_labels")
    except ValueError:
        pass
    # attention_mask (unet_controlnet.py:366-369,421-443): prepared and handed to the blocks, whose forwards never pass it to their
    # transformers (unet_3d_blocks.py:276-283,384-410,618-660; Transformer3DModel.forward has no such parameter) - a DEAD input
    um = load_synth(UNet3DConditionModel(**cases.TINY_MOTION))
    mask = (seeded_randn((2, 256), 72) > 0).float()
    T["attention_mask/out"] = um(x, 961, ctx, attention_mask=mask).sample
    assert torch.equal(T["attention_mask/out"], um(x, 961, ctx).sample), "attention_mask is live in the reference"
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "unet_switches.safetensors"))
    print("unet_switches.safetensors", {k: tuple(v.shape) for k, v in T.items()})


# =============================================================================== loop re-enactment
def gen_loop(u1, ref):
    """Re-enact EMOAnimationPipeline.py:698-823 around the reference's own UNet (the file itself
    cannot be imported: `animated_diff`, diffusers pipelines).  Scheduler = oracle/scheduler_ref.py
    (diffusers absent => parity unpinned there).

    Variants (key prefix): ddim / ddpm = CFG 7.5, context_batch_size 1;  ddim_cbs2 = context_batch_size 2 with THREE windows
    (a full batch of two + a partial batch), text rows exactly as the reference builds them - `torch.cat([text] * cbs)` (:631)
    is [uc, c, uc, c] against latent rows [w0, w1, w0, w1] and the reader's uc mask [1, 1, 0, 0];  ddim_nocfg =
    guidance_scale 1.0: `do_classifier_free_guidance = guidance_scale > 1.0` (:622) is False, the UNet batch is the window
    batch (:759-763 `.repeat(1)`), text is the cond embedding alone (_encode_prompt without the uncond half), eps =
    noise_pred / counter (:812-814 skipped).  The reference's own no-CFG lines do not run as written (`pred_uc, pred_c =
    pred.chunk(2)` at :790 unpacks a one-row batch, and the reader is built with do_classifier_free_guidance=True at :634, which
    would mask the first half of the FRAMES off the bank): the golden re-enacts the evident intent - no chunk, every row reads
    the bank (reader built with do_classifier_free_guidance=False)."""
    import math as _m

    from oracle.scheduler_ref import SchedulerRef, counter_normal
    T = {}

    def run(prefix, kind, cbs=1, gs=7.5, ov=2):
        cfg = gs > 1.0
        sch = SchedulerRef(kind)
        steps = 3
        timesteps = sch.set_timesteps(steps)
        f_tot, cf = 8, 4
        latents = seeded_randn((1, 4, f_tot, 16, 16), 5)
        ref_lat = seeded_randn((1, 4, 16, 16), 3)
        text2 = seeded_randn((2, 5, 32), 2)
        text = torch.cat([text2 if cfg else text2[1:]] * cbs)                                  # :625-631
        nbr = 2 if cfg else 1
        for si, t in enumerate(timesteps):
            noise_pred = torch.zeros(nbr, *latents.shape[1:])
            counter = torch.zeros(1, 1, f_tot, 1, 1)
            banks = run_writer(ref, ref_lat.repeat(nbr * cbs, 1, 1, 1), t, text)               # :711-716
            queue = list(ref_uniform(0, steps, f_tot, cf, 1, ov))                             # :748-750
            nb = _m.ceil(len(queue) / cbs)
            for i in range(nb):
                context = queue[i * cbs:(i + 1) * cbs]
                lmi = torch.cat([latents[:, :, c] for c in context]).repeat(nbr, 1, 1, 1, 1)   # :759-763
                b = lmi.shape[0]
                # (a partial last batch: `.repeat(1, video_length, 1, 1)[:hidden_states.shape[0]]` (:236) cuts the bank rows to the batch)
                pred = run_reader(u1, lmi, t, text[:b], banks, cfg=cfg, batch_size=cbs)       # :774-788
                if cfg:
                    pred_uc, pred_c = pred.chunk(2)
                    pred = torch.cat([pred_uc.unsqueeze(0), pred_c.unsqueeze(0)])
                else:
                    pred = pred.unsqueeze(0)
                for j, c in enumerate(context):
                    noise_pred[:, :, c] = noise_pred[:, :, c] + pred[:, j]                   # :792-794
                    counter[:, :, c] = counter[:, :, c] + 1
            if cfg:
                uc, cc = (noise_pred / counter).chunk(2)                                      # :813
                eps = uc + gs * (cc - uc)                                                     # :814
            else:
                eps = noise_pred / counter
            T[f"{prefix}/eps{si}"] = eps.clone()
            z = counter_normal(0, si, latents.numel()).reshape(latents.shape) if kind == "ddpm" else None
            latents = sch.step(eps, t, latents, noise=z)
        T[f"{prefix}/latents"] = latents

    run("ddim", "ddim")
    run("ddpm", "ddpm")
    run("ddim_cbs2", "ddim", cbs=2)
    run("ddim_nocfg", "ddim", gs=1.0)
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "loop_tiny.safetensors"))
    print("loop_tiny.safetensors", list(T))


# =============================================================================== conditioning (A17/A18)
def gen_conditioning():
    import torch.nn as nn
    T = {}
    N = shim.extract_classes("/root/reference/Net.py",
                             ["SpeedEncoder", "CrossAttentionLayer", "AudioAttentionLayers", "ReferenceAttentionLayer"])
    N2 = shim.extract_classes("/root/reference/Net.py", ["FaceLocator"])
    S2 = shim.extract_classes("/root/reference/train_stage_2_temporal_audio.py", ["TemporalAttention", "AudioAttention"])
    S3 = shim.extract_classes("/root/reference/train_stage_3_speedlayers.py", ["SpeedController", "FaceRegionController"])
    speeds = torch.tensor(cases.SPEEDS, dtype=torch.float32)
    se = load_synth(N["SpeedEncoder"](9, 64), "speed_encoder.")
    T["speed_encoder/encode"] = se.encode_speed(speeds)
    T["speed_encoder/out"] = se(speeds)
    sc = load_synth(S3["SpeedController"](9, 64), "speed_controller.")
    T["speed_controller/out"] = sc(speeds)
    fr = load_synth(S3["FaceRegionController"](1, 32), "face_region.")
    T["face_region/out"] = fr(seeded_randn((2, 1, 8, 8), 60))
    al = load_synth(N["AudioAttentionLayers"](48, 2), "audio_layers.")
    T["audio_layers/out"] = al(seeded_randn((2, 6, 48), 61), seeded_randn((2, 6, 48), 62))
    rl = load_synth(N["ReferenceAttentionLayer"](48), "ref_layer.")
    T["ref_layer/out"] = rl(seeded_randn((2, 6, 48), 63), seeded_randn((2, 1, 48), 64))
    aa = load_synth(S2["AudioAttention"](64, 768, 8), "stage2_audio.")
    T["stage2_audio/out"] = aa(seeded_randn((2, 12, 64), 65), seeded_randn((2, 5, 768), 66))
    ta = load_synth(S2["TemporalAttention"](64, 8), "stage2_temporal.")
    T["stage2_temporal/out"] = ta(seeded_randn((2, 12, 64), 67))
    fl = load_synth(N2["FaceLocator"](), "face_locator.")
    T["face_locator/out"] = fl(seeded_randn((2, 3, 32, 48), 68))
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "conditioning.safetensors"))
    print("conditioning.safetensors", list(T))


# =============================================================================== VideoNet attention modules
def gen_videonet():
    """models/videonet.py:15-128: SpatialAttentionModule (reference features concatenated along the width) and
    TemporalAttentionModule, AST-extracted (the file imports xformers / diffusers at module level).  The one third-party op,
    xformers.ops.memory_efficient_attention, is stood in by its published semantics softmax(q k^T K^-0.5) v for [B, M, H, K] /
    [B, M, K] inputs (no bias, p = 0) - everything else executed is the reference's own code."""
    from einops import rearrange

    def mea(q, k, v):
        if q.dim() == 4:
            q, k, v = (t.transpose(1, 2) for t in (q, k, v))
            return (torch.softmax(q @ k.transpose(-1, -2) * q.shape[-1] ** -0.5, -1) @ v).transpose(1, 2)
        return torch.softmax(q @ k.transpose(-1, -2) * q.shape[-1] ** -0.5, -1) @ v

    V = shim.extract_classes("/root/reference/models/videonet.py", ["SpatialAttentionModule", "TemporalAttentionModule"],
                             extra_ns={"rearrange": rearrange, "memory_efficient_attention": mea})
    T = {}
    sp = load_synth(V["SpatialAttentionModule"](64, embed_dim=64, num_heads=8), "videonet_spatial.")
    x, r = seeded_randn((3, 64, 4, 8), 80), seeded_randn((3, 64, 4, 8), 81)
    T["spatial/out"] = sp(x, r)
    tm = load_synth(V["TemporalAttentionModule"](64, 4, embed_dim=64, num_heads=8), "videonet_temporal.")
    T["temporal/out"] = tm(seeded_randn((2 * 4, 64, 4, 4), 82))
    # ---- ReferenceConditionedAttentionBlock.forward (models/videonet.py:132-196): sam -> cross_attn -> tam, the reference's own
    # class body.  Its two third-party members are stood in by the reference's IN-TREE equivalents: `cross_attn` (a diffusers
    # Transformer2DModel) by magicanimate's Transformer3DModel at one frame behind the 2-D call signature, and
    # models/motionmodule.get_motion_module (diffusers Attention / FeedForward inside) by magicanimate/models/motion_module.py's
    # (same module tree and key names, in-tree attention) - with the ctor's own defaults, i.e. TWO transformer blocks, no PE.
    from types import SimpleNamespace

    class CrossAttn2D(ref_attention.Transformer3DModel):
        def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, added_cond_kwargs=None, class_labels=None,
                    cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None, return_dict=True):
            y = super().forward(hidden_states[:, :, None], encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
            return (y[:, :, 0],)

    R = shim.extract_classes("/root/reference/models/videonet.py", ["SpatialAttentionModule", "ReferenceConditionedAttentionBlock", "VideoNet"],
                             extra_ns={"rearrange": rearrange, "memory_efficient_attention": mea, "Transformer2DModel": CrossAttn2D,
                                       "UNet2DConditionModel": object, "get_motion_module": ref_mm.get_motion_module,
                                       "copy": __import__("copy")})
    C_, heads, nfr = 64, 8, 4
    ca = CrossAttn2D(num_attention_heads=heads, attention_head_dim=C_ // heads, in_channels=C_, cross_attention_dim=32, norm_num_groups=32,
                     unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
    blk = load_synth(R["ReferenceConditionedAttentionBlock"](ca, nfr), "rcab.")
    x, r, ctx = seeded_randn((2 * nfr, C_, 4, 8), 83), seeded_randn((2 * nfr, C_, 4, 8), 84), seeded_randn((2 * nfr, 5, 32), 85)
    blk.update_reference_tensor(r)
    T["rcab/out"] = blk(x, ctx)[0]
    blk.skip_temporal_attn = True
    T["rcab/out_skip"] = blk(x, ctx)[0]
    blk.skip_temporal_attn = False
    blk.update_num_frames(2)          # the same rows regrouped as 4 clips of 2 frames
    T["rcab/out_frames2"] = blk(x, ctx)[0]
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "videonet.safetensors"))
    print("videonet.safetensors", {k: tuple(v.shape) for k, v in T.items()})
    # ---- VideoNet.__init__ / update_reference_embeddings (models/videonet.py:199-247) on the reference's own tiny 3-D UNet (a
    # diffusers 2-D UNet is absent; the ctor only walks down_blocks / mid_block / up_blocks `.attentions`): which attention slots
    # become ReferenceConditionedAttentionBlocks, in which ORDER the reference embeddings are dealt to them, and the resulting
    # state-dict key listing (INT goldens, tests/golden/ints.json is rewritten by gen_ints: stored in their own file here)
    sd_unet = UNet3DConditionModel(**cases.VIDEONET_TINY)
    vn = R["VideoNet"](sd_unet, num_frames=nfr)
    names = {id(m): n for n, m in vn.unet.named_modules()}
    order = [names[id(b)] for b in vn.ref_cond_attn_blocks]
    refs = [torch.full((1,), float(i)) for i in range(len(order))]
    vn.update_reference_embeddings(refs)
    dealt = [int(b.reference_tensor.item()) for b in vn.ref_cond_attn_blocks]
    keys = [[k, list(v.shape)] for k, v in vn.state_dict().items()]
    json.dump({"block_order": order, "reference_index_of_block": dealt, "n_keys": len(keys), "keys": keys},
              open(os.path.join(GOLD, "videonet_wiring.json"), "w"), indent=0)
    print("videonet_wiring.json", len(order), "blocks,", len(keys), "keys")


# =============================================================================== Net.py placeholders
def gen_net_placeholders():
    """Net.py's never-wired placeholder modules, run as they stand (AST-extracted class bodies) wherever they run at all:
      ReferenceAttention  :1487-1511   single-head attention of LN(x) tokens over LN(ref) keys / raw ref values, no residual
      MotionModule + TemporalAttention :1449-1485   Conv3d over time + nn.MultiheadAttention over (T, B, C*H*W) - LayerNorm(channels)
                          is applied to a C*H*W-wide axis, so the pair runs for 1x1 feature maps (and odd temporal_length) only
      TemporalModule      :520-552     `x.view(b, -1, c, h, w)` hands Conv3d a 1-channel volume: raises for every channels % 8 == 0
      BackboneNetwork     :368-411     reference- and audio-attention stacks run; the VanillaTemporalModule stage receives the 3-D
                          latent and trips its own ndim == 5 assertion
    What runs is stored as tensors; what does not is stored as the exception type the reference raises (net_placeholders.json)."""
    T, J = {}, {}
    N = shim.extract_classes("/root/reference/Net.py", ["ReferenceAttention", "MotionModule", "TemporalAttention", "TemporalModule",
                                                        "BackboneNetwork", "ReferenceAttentionLayer", "AudioAttentionLayers", "CrossAttentionLayer"],
                             extra_ns={"VanillaTemporalModule": ref_mm.VanillaTemporalModule})
    ra = load_synth(N["ReferenceAttention"](64), "net_reference_attention.")
    T["reference_attention/out"] = ra(seeded_randn((2, 64, 4, 6), 400), seeded_randn((2, 64, 4, 6), 401))
    mm = load_synth(N["MotionModule"](64, 3), "net_motion_module.")
    T["motion_module/out"] = mm(seeded_randn((2, 64, 6, 1, 1), 402))
    for name, fn in (("motion_module_4x4", lambda: mm(seeded_randn((2, 64, 6, 4, 4), 403))),
                     ("motion_module_even_kernel", lambda: load_synth(N["MotionModule"](64, 4), "net_motion_module4.")(seeded_randn((2, 64, 6, 1, 1), 402))),
                     ("temporal_module", lambda: load_synth(N["TemporalModule"](64, 4), "net_temporal_module.")(seeded_randn((2, 64, 4, 4), 404), None))):
        try:
            fn()
            J[name] = "runs"
        except Exception as ex:
            J[name] = type(ex).__name__
    feat = 32
    audio_layers = load_synth(N["AudioAttentionLayers"](feat, 2), "net_backbone_audio.")
    bb = N["BackboneNetwork"](feat, 2, lambda img: img, audio_layers, temporal_module_kwargs=dict(num_attention_heads=4, num_transformer_block=1))
    load_synth(bb.reference_attention_layers, "net_backbone_ref.")
    lat, aud, ref = seeded_randn((2, 5, feat), 405), seeded_randn((2, 5, feat), 406), seeded_randn((2, 1, feat), 407)
    try:
        bb(lat, aud, ref)
        J["backbone_forward"] = "runs"
    except Exception as ex:
        J["backbone_forward"] = type(ex).__name__
    x = lat
    for layer in bb.reference_attention_layers:                    # the stages in front of the assertion, as :401-407 run them
        x = layer(x, ref) + x
    T["backbone/before_temporal"] = bb.audio_attention_layers(x, aud)
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "net_placeholders.safetensors"))
    json.dump(J, open(os.path.join(GOLD, "net_placeholders.json"), "w"), indent=1, sort_keys=True)
    print("net_placeholders", {k: tuple(v.shape) for k, v in T.items()}, J)


# =============================================================================== EMOAnimationPipeline methods next to __call__
def _pipeline_methods(names):
    """AST-extract methods of the EMOAnimationPipeline class (the file does not import here) as plain functions taking `self`."""
    import ast
    import typing

    import numpy as np
    from einops import rearrange
    util = ast.parse(open("/root/reference/magicanimate/utils/util.py").read())
    ns = {"torch": torch, "np": np, "rearrange": rearrange, "tqdm": lambda it, **kw: it, "Optional": typing.Optional, "List": typing.List,
          "Union": typing.Union, "Callable": typing.Callable}
    keep = [n for n in util.body if (isinstance(n, ast.FunctionDef) and n.name in ("get_tensor_interpolation_method", "set_tensor_interpolation_method",
                                                                                   "linear", "slerp"))
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "tensor_interpolation")]
    exec(compile(ast.Module(body=keep, type_ignores=[]), "util.py", "exec"), ns)
    tree = ast.parse(open("/root/reference/EMOAnimationPipeline.py").read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "EMOAnimationPipeline")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    exec(compile(ast.Module(body=fns, type_ignores=[]), "EMOAnimationPipeline.py", "exec"), ns)
    return ns


def gen_pipeline_methods(u1):
    """The methods of EMOAnimationPipeline around its __call__ (EMOAnimationPipeline.py:341-540), run as they stand on stub `self` objects:
    prepare_latents, prepare_condition, next_step, invert (around the reference's own tiny UNet; tokenizer / text encoder / VAE stubbed by
    tensors), interpolate_latents (+ magicanimate/utils/util.py slerp / linear), select_controlnet_res_samples."""
    import numpy as np
    from types import SimpleNamespace

    from oracle.scheduler_ref import SchedulerRef
    M = _pipeline_methods(["prepare_latents", "prepare_condition", "next_step", "invert", "interpolate_latents", "select_controlnet_res_samples"])
    T = {}

    def sched(n):
        r = SchedulerRef("ddim")
        r.set_timesteps(n)
        return SimpleNamespace(config=SimpleNamespace(num_train_timesteps=1000), num_inference_steps=n, alphas_cumprod=r.alphas_cumprod,
                               final_alpha_cumprod=r.final_alpha_cumprod, timesteps=list(r.timesteps), init_noise_sigma=1.0,
                               set_timesteps=lambda k: None)
    me = SimpleNamespace(scheduler=sched(50), vae_scale_factor=8)
    x, eps = seeded_randn((4, 4, 16, 16), 500), seeded_randn((4, 4, 16, 16), 501)
    for t in (1, 21, 481, 981):
        xn, x0 = M["next_step"](me, eps, t, x)
        T[f"next_step/{t}/x_next"], T[f"next_step/{t}/pred_x0"] = xn, x0
    T["prepare_latents/out"] = M["prepare_latents"](me, 1, 4, 32, 64, 64, torch.float32, torch.device("cpu"), torch.Generator().manual_seed(5))
    cond = np.random.RandomState(7).randint(0, 256, size=(3, 16, 24, 3)).astype(np.uint8)
    T["prepare_condition/in"] = torch.from_numpy(cond.copy())
    T["prepare_condition/out"] = M["prepare_condition"](me, cond, 1, "cpu", torch.float32, True)
    lat = seeded_randn((1, 4, 3, 4, 4), 502)
    for name, is_slerp in (("slerp", True), ("linear", False)):
        M["set_tensor_interpolation_method"](is_slerp)
        T[f"interpolate/{name}"] = M["interpolate_latents"](me, lat, 3, "cpu")
    cache = {i: ([seeded_randn((1, 8, 4, 4), 600 + 10 * i + k) for k in range(3)], seeded_randn((1, 8, 2, 2), 700 + i)) for i in range(6)}
    down, mid = M["select_controlnet_res_samples"](me, cache, [[0, 1], [4, 5]], True, 4, 2)
    for k, d in enumerate(down):
        T[f"select/down{k}"] = d
    T["select/mid"] = mid
    # invert (:417-477) around the reference's own tiny motion UNet: 5 inversion steps scheduled, 3 taken, 4 frames
    text = seeded_randn((1, 5, 32), 503)
    frames = seeded_randn((4, 4, 16, 16), 504)
    inv = SimpleNamespace(scheduler=sched(5), unet=u1, _execution_device="cpu",
                          tokenizer=lambda prompt, **kw: SimpleNamespace(input_ids=torch.zeros(1, 77, dtype=torch.long)),
                          text_encoder=lambda ids: [text], images2latents=lambda image: image)
    inv.next_step = lambda *a, **k: M["next_step"](inv, *a, **k)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        out, inter = M["invert"](inv, frames, "", num_inference_steps=5, num_actual_inference_steps=3, return_intermediates=True)
    T["invert/latents"] = out
    T["invert/step1"] = inter[1]
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(GOLD, "pipeline_methods.safetensors"))
    print("pipeline_methods.safetensors", {k: tuple(v.shape) for k, v in T.items() if "/" in k and not k.startswith("next_step")})


# =============================================================================== audio windows
def gen_audio_windows():
    """Net.py:649-667: the per-frame windowing loop of Wav2VecFeatureExtractor.extract_features_from_wav, run on synthetic
    `hidden_states` (the class itself needs transformers checkpoints + soundfile / librosa).  The statements from
    `num_frames = hidden_states.shape[1]` to `all_features = torch.stack(...)` are taken from the method body by AST and
    executed as they stand; only tensors are stored."""
    import ast
    from types import SimpleNamespace
    tree = ast.parse(open("/root/reference/Net.py").read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Wav2VecFeatureExtractor")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "extract_features_from_wav")
    start = next(i for i, st in enumerate(fn.body) if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "num_frames")
    stop = next(i for i, st in enumerate(fn.body) if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "all_features"
                and isinstance(st.value, ast.Call) and getattr(st.value.func, "attr", "") == "stack")
    code = compile(ast.Module(body=fn.body[start:stop + 1], type_ignores=[]), "Net.py", "exec")
    T = {}
    for name, (Tn, D, m, n) in {"t9_m2n2": (9, 16, 2, 2), "t3_m2n2": (3, 16, 2, 2), "t7_m1n3": (7, 8, 1, 3), "t5_m0n0": (5, 8, 0, 0)}.items():
        hs = seeded_randn((1, Tn, D), 300 + Tn)
        ns = dict(torch=torch, hidden_states=hs, m=m, n=n, self=SimpleNamespace(device="cpu"))
        exec(code, ns)
        T[f"{name}/in"] = hs[0].contiguous()
        T[f"{name}/out"] = ns["all_features"].contiguous()
    save_file(T, os.path.join(GOLD, "audio_windows.safetensors"))
    print("audio_windows.safetensors", {k: tuple(v.shape) for k, v in T.items()})


# =============================================================================== ControlNet
def gen_controlnet():
    """The in-tree arithmetic of magicanimate/models/controlnet.py that does not need diffusers: the
    ControlNetConditioningEmbedding class (AST-extracted, :49-91) with synthesised weights under the module's own key names
    (prefix controlnet_cond_embedding.) on a 3x64x64 conditioning image."""
    import torch.nn as nn

    def zero_module(module):   # controlnet.py:570-573 (the weights are overwritten by load_synth anyway)
        for p_ in module.parameters():
            nn.init.zeros_(p_)
        return module

    C = shim.extract_classes("/root/reference/magicanimate/models/controlnet.py", ["ControlNetConditioningEmbedding"],
                             extra_ns={"zero_module": zero_module})
    ce = load_synth(C["ControlNetConditioningEmbedding"](conditioning_embedding_channels=32, block_out_channels=(16, 32, 96, 256)),
                    "controlnet_cond_embedding.")
    y = ce(seeded_randn((2, 3, 64, 64), 70))
    save_file({"cond_embedding/out": y.contiguous()}, os.path.join(GOLD, "controlnet.safetensors"))
    print("controlnet.safetensors", tuple(y.shape), float(y.abs().mean()))


# =============================================================================== cfg1 (BASELINE config 1)
def gen_cfg1():
    """BASELINE.json configs[0]: UNet built literally from configs/unet-config.yaml:default
    (norm_num_groups: 4), (1,4,1,32,32), timestep 981, ctx (1,77,768), no motion, random weights."""
    from emote_hack_amd.config import unet_config_from_yaml
    cfg = unet_config_from_yaml("/root/reference/configs/unet-config.yaml")
    t0 = time.time()
    u = load_synth(UNet3DConditionModel(**{k: v for k, v in cfg.items() if k != "motion_module_kwargs"}))
    x = seeded_randn((1, 4, 1, 32, 32), 1)
    ctx = seeded_randn((1, 77, 768), 2)
    y = u(x, 981, ctx).sample
    save_file({"cfg1/out": y.contiguous()}, os.path.join(GOLD, "cfg1.safetensors"))
    print("cfg1.safetensors", tuple(y.shape), float(y.abs().mean()), f"{time.time() - t0:.1f}s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-cfg1", action="store_true")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["ints", "modules", "unet", "cond", "controlnet", "audio", "videonet", "net", "switches", "cfg1"]
    if "ints" in todo:
        gen_ints()
    if "modules" in todo:
        gen_modules()
    if "unet" in todo:
        u1, ref = gen_unet_tiny()
        gen_loop(u1, ref)
        gen_pipeline_methods(u1)
    if "cond" in todo:
        gen_conditioning()
    if "controlnet" in todo:
        gen_controlnet()
    if "audio" in todo:
        gen_audio_windows()
    if "videonet" in todo:
        gen_videonet()
    if "net" in todo:
        gen_net_placeholders()
    if "switches" in todo:
        gen_unet_switches()
    if "cfg1" in todo and not a.skip_cfg1:
        gen_cfg1()
