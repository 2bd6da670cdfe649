#!/usr/bin/env python3
"""Where does the low-precision error of the UNet come from?  

Teacher-forced LOCAL error per sub-block (resnet / transformer / motion module) on the tiny motion UNet: every block of
the low-precision model is fed the f32 model's input for that block (rounded to the compute dtype), so its output error
against the f32 block output is the error this block ADDS, not what it inherited.

    --oracle : CPU, oracle/unet_ref.py run in torch bf16 (= what the reference's own bf16 forward does per block)
    --hip    : GPU, the HIP kernels in bf16 (or --dtype f16)

Printing both side by side shows which block kind of the HIP path is worse than a plain torch bf16 implementation.
Test infrastructure (it imports oracle/): never imported by the product."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from emote_hack_amd.spec import build_spec, param_shapes  # noqa: E402
from emote_hack_amd.synth import synth_state_dict  # noqa: E402
from tests import cases  # noqa: E402


def report(rows, title):
    print(f"== {title}")
    print(f"{'block':58s} {'mean|y|':>9s} {'mean err':>9s} {'max err':>9s} {'rel mean':>9s}")
    by_kind = {}
    for kind, name, ref, got in rows:
        e = (got - ref).abs()
        ma, me, mx = float(ref.abs().mean()), float(e.mean()), float(e.max())
        print(f"{kind + ' ' + name:58s} {ma:9.4f} {me:9.2e} {mx:9.2e} {me / ma:9.2e}")
        by_kind.setdefault(kind, []).append(me / ma)
    for k, v in by_kind.items():
        print(f"   {k:12s} mean relative local error {sum(v) / len(v):.3e}  (worst {max(v):.3e}, n={len(v)})")


def run_oracle(cfg, x, ctx, t, lowp):
    from oracle import unet_ref as U
    sd = synth_state_dict(param_shapes(build_spec(cfg)))
    rec = []
    orig = {n: getattr(U, n) for n in ("resnet_block", "transformer3d", "motion_module")}

    def wrap(name):
        def f(sd_, p, *a, **kw):
            y = orig[name](sd_, p, *a, **kw)
            rec.append((name, p, a, kw, y))
            return y
        return f
    for n in orig:
        setattr(U, n, wrap(n))
    y32 = U.unet_forward(sd, cfg, x, t, ctx)
    for n in orig:
        setattr(U, n, orig[n])
    sd16 = {k: v.to(lowp) for k, v in sd.items()}
    cast = lambda v: v.to(lowp) if torch.is_tensor(v) and v.is_floating_point() else v
    rows = []
    for name, p, a, kw, y in rec:
        y16 = orig[name](sd16, p, *[cast(v) for v in a], **{k: cast(v) for k, v in kw.items()})
        rows.append((name, p, y, y16.float()))
    report(rows, f"oracle in torch {lowp} on CPU (local error per block)")
    # end to end: the whole model in low precision
    try:
        te = U.timestep_embedding
        U.timestep_embedding = lambda *a_, **k_: te(*a_, **k_).to(lowp)   # unet_controlnet.py:397 `t_emb.to(dtype=self.dtype)`
        y16 = U.unet_forward(sd16, cfg, x.to(lowp), t, ctx.to(lowp)).float()
        U.timestep_embedding = te
        e = (y16 - y32).abs()
        print(f"end to end: mean|y| {float(y32.abs().mean()):.4f}  mean err {float(e.mean()):.3e}  max err {float(e.max()):.3e}")
    except Exception as ex:   # the oracle's f32 timestep table etc. may refuse a low-precision end-to-end run
        print("end to end oracle run in low precision failed:", ex)


def run_hip(cfg, x, ctx, t, lowp):
    from emote_hack_amd.unet import UNet3DConditionModel
    sd = synth_state_dict(param_shapes(build_spec(cfg)))
    dev = "cuda"

    def build(dtype):
        m = UNet3DConditionModel(**cfg)
        m.load_state_dict(sd)
        return m.to(dev, dtype)
    m32, m16 = build(torch.float32), build(lowp)
    rec = []
    names = ("_resnet", "_transformer", "_motion")
    orig32 = {n: getattr(m32, n) for n in names}

    def wrap32(name):
        def f(spec, x_, *a, **kw):
            xin = x_.float().clone()
            y = orig32[name](spec, x_, *a, **kw)
            rec.append((name, spec.prefix, xin, y.float().clone()))
            return y
        return f
    for n in names:
        setattr(m32, n, wrap32(n))
    y32 = m32(x.to(dev), t, ctx.to(dev)).sample.float().cpu()
    rows, it = [], iter(rec)
    orig16 = {n: getattr(m16, n) for n in names}

    def wrap16(name):
        def f(spec, x_, *a, **kw):
            kind, prefix, xin, yref = next(it)
            assert kind == name and prefix == spec.prefix
            y = orig16[name](spec, xin.to(lowp), *a, **kw)       # teacher forcing: the f32 model's input, rounded
            rows.append((name, prefix, yref.cpu(), y.float().cpu()))
            return y
        return f
    for n in names:
        setattr(m16, n, wrap16(n))
    m16(x.to(dev), t, ctx.to(dev))
    report(rows, f"HIP kernels in {lowp} (teacher-forced local error per block)")
    for n in names:
        setattr(m16, n, orig16[n])
    y16 = m16(x.to(dev), t, ctx.to(dev)).sample.float().cpu()
    e = (y16 - y32).abs()
    print(f"end to end: mean|y| {float(y32.abs().mean()):.4f}  mean err {float(e.mean()):.3e}  max err {float(e.max()):.3e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--config", default="tiny", choices=["tiny", "medium"])
    a = ap.parse_args()
    lowp = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    torch.set_grad_enabled(False)
    if a.config == "tiny":
        cfg = cases.TINY_MOTION
        x, ctx = cases.tiny_inputs(2, 4)
    else:   # SD-1.5 widths, two levels, 16x16 latent, F=3 (real head dims 40 / 80)
        from emote_hack_amd.synth import seeded_randn
        cfg = dict(cases.SD15_MOTION, block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                   up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), attention_head_dim=8, layers_per_block=1)
        x, ctx = seeded_randn((2, 4, 3, 16, 16), 1), seeded_randn((2, 9, 768), 2)
    if a.oracle:
        run_oracle(cfg, x, ctx, 961, lowp)
    if a.hip:
        run_hip(cfg, x, ctx, 961, lowp)
