"""GPU parity at BASELINE.json's FULL sizes, through size-independent properties (the CPU oracle needs minutes per forward
at these sizes; the f32 HIP path is pinned to the oracle / reference goldens at the small sizes of test_gpu_unet.py):

  configs[4] ("cfg5"): the 4-level SD-1.5 network + motion modules at 768x768 (96x96 latents), a 24-frame window, fp16,
      speed-layer embeddings AND per-frame audio context (48, 5, 768) - fp16 HIP vs f32 HIP within the reference's own fp16
      error scale, CFG batch rows independent of each other.
  configs[1]/[2] ("cfg2"/"cfg3") with the ReferenceNet ON: full-size write pass -> projected banks -> sampling-loop steps;
      eager launches vs HIP-graph replay bit-identical, ReferenceNet timesteps batched 1 / 3 per pass equal within the
      bf16 error scale, bf16 vs f32 eps of the first step within the reference's own bf16 error scale.
"""
import pytest
import torch

from emote_hack_amd.synth import seeded_randn, synth_state_dict
from tests import cases
from tests.test_gpu_unet import check

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(cfg, dtype, prefix="", cls=None):
    from emote_hack_amd.spec import param_shapes
    from emote_hack_amd.unet import UNet3DConditionModel
    m = (cls or UNet3DConditionModel)(**cfg)
    m.load_state_dict(synth_state_dict(param_shapes(m.spec), prefix=prefix, device=DEV))   # device generator: seconds, not minutes
    return m.to(DEV, dtype)


def test_cfg5_full_network_fp16_speed_and_audio():
    x = seeded_randn((2, 4, 24, 96, 96), 1)
    ctx = seeded_randn((2, 77, 768), 2)
    audio = seeded_randn((2 * 24, 5, 768), 4)
    speed = 0.1 * seeded_randn((2, 4 * 320), 5)
    kw = lambda sl=slice(None): dict(audio_features=audio.reshape(2, 24, 5, 768)[sl].reshape(-1, 5, 768).to(DEV),
                                     speed_embeddings=speed[sl].to(DEV))
    torch.manual_seed(0)
    m32 = build(cases.SD15_MOTION, torch.float32)
    sd = {k: v for k, v in m32._master.items()}
    y32 = m32(x.to(DEV), 500, ctx.to(DEV), **kw()).sample.float().cpu()
    del m32
    torch.cuda.empty_cache()
    from emote_hack_amd.unet import UNet3DConditionModel
    m16 = UNet3DConditionModel(**cases.SD15_MOTION)
    m16.load_state_dict(sd)
    m16.to(DEV, torch.float16)
    y16 = m16(x.to(DEV), 500, ctx.to(DEV), **kw()).sample
    assert y16.shape == (2, 4, 24, 96, 96) and bool(torch.isfinite(y16).all())
    check(y16, y32, torch.float16)
    # batch rows are independent (GroupNorm is per batch row): the cond row alone reproduces row 1 of the batched call
    y1 = m16(x[1:].to(DEV), 500, ctx[1:].to(DEV), **kw(slice(1, 2))).sample
    torch.testing.assert_close(y1.float().cpu(), y16[1:].float().cpu(), rtol=2e-2, atol=2e-2)
    # the audio context and the speed embedding are live inputs
    y_na = m16(x.to(DEV), 500, ctx.to(DEV), speed_embeddings=speed.to(DEV)).sample
    y_ns = m16(x.to(DEV), 500, ctx.to(DEV), audio_features=kw()["audio_features"]).sample
    assert float((y_na.float() - y16.float()).abs().mean()) > 1e-3 and float((y_ns.float() - y16.float()).abs().mean()) > 1e-4


@pytest.fixture(scope="module")
def cfg2_models():
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    torch.manual_seed(0)
    unet = build(cases.SD15_MOTION, torch.bfloat16)
    ref = build(cases.SD15, torch.bfloat16, cases.REF_PREFIX, cls=AppearanceEncoderModel)
    return unet, ref


def _run_loop(unet, ref, steps, *, graphs, ref_group, audio=None, n_steps=50, frames=12, ddim=False, frame_slice=None):
    from emote_hack_amd import DDIMScheduler, DDPMScheduler
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDIMScheduler() if ddim else DDPMScheduler())
    lat0 = seeded_randn((1, 4, frames, 64, 64), 1)
    if frame_slice is not None:
        lat0 = lat0[:, :, frame_slice].contiguous()
    st = pipe.prepare_denoise(lat0.to(DEV), seeded_randn((1, 4, 64, 64), 3), seeded_randn((2, 77, 768), 2),
                              appearance_encoder=ref, num_inference_steps=n_steps, guidance_scale=7.5, context_frames=12,
                              context_stride=1, context_overlap=0, seed=0, use_graphs=graphs, reference_group=ref_group,
                              return_eps=True, audio_features=audio)
    for si in range(steps):
        pipe.denoise_step(st, si)
    torch.cuda.synchronize()
    return st.latents.clone(), [e.clone() for e in st.eps_trace]


def test_cfg2_reference_net_on_eager_vs_graph_replay_bit_identical(cfg2_models):
    """4 steps of the full-size loop (ReferenceNet write pass, projected banks with Lk = 4096 + 4096 at d = 40, cond-only bank
    segment, fused sampler step): HIP-graph replay (steps 2, 3 are replays) must reproduce the eager launches BIT for bit -
    same kernels, same order, same addresses-independent arithmetic."""
    unet, ref = cfg2_models
    lat_e, eps_e = _run_loop(unet, ref, 4, graphs=False, ref_group=3)
    lat_g, eps_g = _run_loop(unet, ref, 4, graphs=True, ref_group=3)
    assert bool(torch.isfinite(lat_e).all())
    for a, b in zip(eps_e, eps_g):
        assert torch.equal(a, b)
    assert torch.equal(lat_e, lat_g)


@pytest.fixture(scope="module")
def cfg2_models_f32(cfg2_models):
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.unet import UNet3DConditionModel
    unet, ref = cfg2_models
    u32 = UNet3DConditionModel(**cases.SD15_MOTION)
    u32.load_state_dict(unet._master)
    u32.to(DEV, torch.float32)
    r32 = AppearanceEncoderModel(**cases.SD15)
    r32.load_state_dict(ref._master)
    r32.to(DEV, torch.float32)
    return u32, r32


def test_cfg2_reference_group_size_does_not_change_the_result(cfg2_models_f32):
    """ReferenceNet timesteps batched 3 per pass vs the reference's one pass per step (and the second HBM slot / the row
    index / the look-ahead stream that come with it), in f32 where a wrong bank row would show: the same eps at north_star's
    tolerance (only the summation order of the larger-M GEMM tiles differs)."""
    u32, r32 = cfg2_models_f32
    _, eps_1 = _run_loop(u32, r32, 3, graphs=False, ref_group=1)
    _, eps_3 = _run_loop(u32, r32, 3, graphs=False, ref_group=2)
    for a, b in zip(eps_1, eps_3):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4)


def test_cfg2_reference_net_on_bf16_vs_f32(cfg2_models, cfg2_models_f32):
    """eps of the first step, bf16 HIP vs f32 HIP, ReferenceNet on, CFG 7.5: within 3 x the reference's own bf16 error scale relative
    to the guided eps (the f32 HIP path itself is held to the ORACLE at this size by test_cfg2_full_unet_f32_vs_oracle below)."""
    unet, ref = cfg2_models
    u32, r32 = cfg2_models_f32
    _, eps_b = _run_loop(unet, ref, 1, graphs=False, ref_group=1)
    _, eps_f = _run_loop(u32, r32, 1, graphs=False, ref_group=1)
    from tests.test_gpu_unet import yardstick
    k_mean, _ = yardstick(torch.bfloat16)
    e = (eps_b[0] - eps_f[0]).abs()
    print("cfg2 eps bf16 vs f32: mean err", float(e.mean()), "mean |eps|", float(eps_f[0].abs().mean()))
    # measured: 1.99 x the yard-stick relative to mean |eps| (the two branches' errors add up under eps = 8.5 c - 7.5 uc, but eps itself is
    # that much larger than a branch's output); the worst-case bound of the formula, 16 x, would hide a broken branch
    assert float(e.mean()) <= 3.0 * k_mean * float(eps_f[0].abs().mean()), (float(e.mean()), float(eps_f[0].abs().mean()))


def test_cfg3_audio_context_in_the_loop(cfg2_models):
    """BASELINE configs[2]: cfg2 with per-frame wav2vec audio tokens (12, 5, 768) as the attn2 context of the cond units (uncond
    units get a zero context): runs through graphs, is finite, and differs from the text-context run."""
    unet, ref = cfg2_models
    audio = seeded_randn((12, 5, 768), 4)
    lat_a, eps_a = _run_loop(unet, ref, 3, graphs=True, ref_group=3, audio=audio)
    lat_t, eps_t = _run_loop(unet, ref, 3, graphs=True, ref_group=3)
    assert bool(torch.isfinite(lat_a).all())
    assert float((eps_a[0] - eps_t[0]).abs().mean()) > 1e-3


def test_cfg4_48_frames_four_windows_full_size(cfg2_models):
    """BASELINE configs[3] ("cfg4") on ONE GPU at full size: a 48-frame clip = 4 windows of 12 at context_overlap 0 x 2 CFG
    branches = 8 units (on 8 GPUs: one per rank, tests/test_dist_gloo.py; here 4 UNet calls of [uc, c] per step), bf16,
    ReferenceNet on (EMOAnimationPipeline.py:752-757,796-821).
      * 50-step DDPM, 2 steps: HIP-graph replay reproduces the eager launches bit for bit;
      * windows are independent at overlap 0 (every frame belongs to one window, counter == 1): the first step's eps of window 0
        equals the eps of the 12-frame single-window run on the same frames, and under the deterministic DDIM sampler so do the
        latents after two steps (the DDPM noise is keyed by the element index of the whole clip, so it differs by construction)."""
    unet, ref = cfg2_models
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    from emote_hack_amd import DDPMScheduler
    st = EMOAnimationPipeline(unet=unet, scheduler=DDPMScheduler()).prepare_denoise(
        seeded_randn((1, 4, 48, 64, 64), 1).to(DEV), seeded_randn((1, 4, 64, 64), 3), seeded_randn((2, 77, 768), 2), appearance_encoder=ref,
        num_inference_steps=50, context_frames=12, context_overlap=0)
    assert [c.units for c in st.calls] == [[(w, 0), (w, 1)] for w in range(4)] and all(c.halves_identical for c in st.calls)
    lat_e, eps_e = _run_loop(unet, ref, 2, graphs=False, ref_group=2, frames=48)
    lat_g, eps_g = _run_loop(unet, ref, 2, graphs=True, ref_group=2, frames=48)
    assert lat_e.shape == (1, 4, 48, 64, 64) and bool(torch.isfinite(lat_e).all())
    for a, b in zip(eps_e, eps_g):
        assert torch.equal(a, b)
    assert torch.equal(lat_e, lat_g)
    _, eps_1 = _run_loop(unet, ref, 1, graphs=False, ref_group=2, frames=48, frame_slice=slice(0, 12))
    assert torch.equal(eps_e[0][:, :, :12], eps_1[0])
    assert not torch.equal(eps_e[0][:, :, 12:24], eps_1[0])               # the other windows see other latents
    lat_48, _ = _run_loop(unet, ref, 2, graphs=True, ref_group=2, frames=48, ddim=True)
    lat_12, _ = _run_loop(unet, ref, 2, graphs=True, ref_group=2, frames=48, ddim=True, frame_slice=slice(36, 48))
    assert torch.equal(lat_48[:, :, 36:], lat_12)                          # the LAST window: the fourth UNet call of a step


@pytest.mark.parametrize("dtype,Fr,H,W", [(torch.float32, 12, 64, 64), (torch.bfloat16, 12, 64, 64),
                                          (torch.float32, 24, 96, 96), (torch.float16, 24, 96, 96)])     # cfg2 / cfg3 geometry; cfg5: 768^2, 24 frames, fp16
def test_level0_transformer_and_motion_block_at_bench_size_vs_oracle(dtype, Fr, H, W):
    """One level-0 Transformer3DModel block (attention.py:112-161,276-320) + its motion module (motion_module.py:139-334) at the size
    bench.py runs them - 12 frames x 64 x 64 tokens x 320 channels, 8 heads of 40, 77 text keys (BASELINE configs[1] / [2]), and
    24 frames x 96 x 96 (configs[4]: 9216 keys per frame, the temporal kernel's two 16-frame blocks) - against the reference-pinned CPU
    oracle (oracle/unet_ref.py), not HIP-vs-HIP: f32 at north_star's rtol 1e-3 / atol 1e-4; bf16 (the run that takes the
    full-size-only plans: 256x256 ping-pong tiles, GroupNorm folded into per-frame proj_in slabs at HW >= 4096, LayerNorm-folded
    projections, the fused ff.net.2 + proj_out tail, the pipelined d = 40 attention) within the low-precision yard-stick."""
    from oracle import unet_ref as U
    from emote_hack_amd import ops
    from emote_hack_amd.spec import build_spec, param_shapes
    from emote_hack_amd.unet import UNet3DConditionModel, _Ctx
    # the first down level only: same level-0 blocks, a fraction of the weights
    cfg = dict(cases.SD15_MOTION, block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
               up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), layers_per_block=1)
    m = UNet3DConditionModel(**cfg)
    sd = synth_state_dict(param_shapes(m.spec))
    m.load_state_dict(sd)
    m.to(DEV, dtype)
    a, mo = m.spec.down[0].attentions[0], m.spec.down[0].motions[0]
    assert a.channels == 320 and a.heads == 8 and mo is not None
    x = seeded_randn((1, 320, Fr, H, W), 21)
    ctx = seeded_randn((1, 77, 768), 22)
    # ---- oracle, frame by frame for the spatial transformer (per-frame GroupNorm, per-frame attention: 2 x 8 x 4096^2 f32 scores a go)
    with torch.no_grad():
        parts = [U.transformer3d(sd, a.prefix, x[:, :, f0:f0 + 2], ctx, 8, 32) for f0 in range(0, Fr, 2)]
        ref_t = torch.cat(parts, 2)
        ref_m = U.motion_module(sd, mo.prefix, ref_t, 8)
    # ---- HIP
    c = _Ctx(1, Fr, H, W)
    rows = ops.ncfhw_to_rows(x.to(DEV), dtype)
    ctx_rows = ops.convert(ctx.to(DEV).float().reshape(-1, 768), dtype)
    yt = m._transformer(a, rows, ctx_rows, 77, Fr, c, H, W)
    got_t = ops.rows_to_ncfhw(yt, 1, 320, Fr, H, W)
    check(got_t, ref_t, dtype)
    # the motion module on the ORACLE's transformer output: each block is pinned on its own
    rows_m = ops.ncfhw_to_rows(ref_t.to(DEV), dtype)
    got_m = ops.rows_to_ncfhw(m._motion(mo, rows_m, c, H, W), 1, 320, Fr, H, W)
    check(got_m, ref_m, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,HW,cin", [(640, 32, 1280), (1280, 16, 2560), (1280, 8, 2560)])
def test_deeper_levels_resnet_transformer_motion_at_bench_size_vs_oracle(dtype, C, HW, cin):
    """The blocks of the 32x32 / 16x16 / 8x8 levels at the size bench.py runs them (12 frames; 640 / 1280 channels, 8 heads of 80 / 160),
    each against the reference-pinned CPU oracle on its own: an UP-path ResnetBlock3D with the concatenated [hidden | skip] input
    (resnet.py:177-207; Cin 1280 / 2560 -> the long-K 3x3 convs: 16-row halo patches at the 32x32 / 16x16 levels, the im2col loader with
    256x256 tiles split 8 ways at 8x8; 1x1 shortcut; temb row bias; joint 5-D GroupNorm), the Transformer3DModel block (merged q | k | v
    launch, d = 80 / 160 attention, GEGLU feed-forward with the fused tail) and its motion module - the full-size-only plans of these
    levels, which the level-0 test above does not reach."""
    from oracle import unet_ref as U
    from emote_hack_amd import ops
    from emote_hack_amd.spec import param_shapes
    from emote_hack_amd.unet import UNet3DConditionModel, _Ctx
    # a two-level network whose FIRST up block has the level's geometry: resnets (cin -> C) + transformers + motion modules at C channels
    cfg = dict(cases.SD15_MOTION, block_out_channels=(C, cin - C), down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
               up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), layers_per_block=1)
    m = UNet3DConditionModel(**cfg)
    sd = synth_state_dict(param_shapes(m.spec))
    m.load_state_dict(sd)
    m.to(DEV, dtype)
    blk = m.spec.up[1]
    r, a, mo = blk.resnets[0], blk.attentions[0], blk.motions[0]
    assert (r.cin, r.cout, a.channels, a.heads) == (cin, C, C, 8) and mo is not None and r.has_shortcut
    Fr, B = 12, 1
    x = seeded_randn((B, cin, Fr, HW, HW), 31)
    ctx = seeded_randn((B, 77, 768), 32)
    emb = seeded_randn((B, 4 * C), 33)
    G, eps = cfg["norm_num_groups"], 1e-5
    with torch.no_grad():
        ref_r = U.resnet_block(sd, r.prefix, x, emb, G, eps)
        ref_t = U.transformer3d(sd, a.prefix, ref_r, ctx, 8, G)
        ref_m = U.motion_module(sd, mo.prefix, ref_t, 8)
    c = _Ctx(B, Fr, HW, HW)
    # the resnet reads its slice of the fused time_emb_proj table (unet._temb_off): F.silu(emb) W^T + b for every resnet of the model
    temb_all = ops.convert(ops.gemm(ops.silu(ops.convert(emb.to(DEV), dtype)), m._w["temb_all.w"], m._w["temb_all.b"]), torch.float32)
    got_r = ops.rows_to_ncfhw(m._resnet(r, ops.ncfhw_to_rows(x.to(DEV), dtype), temb_all, c, HW, HW), B, C, Fr, HW, HW)
    check(got_r, ref_r, dtype)
    ctx_rows = ops.convert(ctx.to(DEV).float().reshape(-1, 768), dtype)
    got_t = ops.rows_to_ncfhw(m._transformer(a, ops.ncfhw_to_rows(ref_r.to(DEV), dtype), ctx_rows, 77, Fr, c, HW, HW), B, C, Fr, HW, HW)
    check(got_t, ref_t, dtype)
    got_m = ops.rows_to_ncfhw(m._motion(mo, ops.ncfhw_to_rows(ref_t.to(DEV), dtype), c, HW, HW), B, C, Fr, HW, HW)
    check(got_m, ref_m, dtype)


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] END TO END against the ORACLE at the benchmarked size: the whole 4-level Backbone (skip connections, the fused
# time_emb_proj table, zero-copy concatenation, the shared CFG prefix) and the ReferenceNet write -> fp16 banks -> read path, not
# block by block.  ~80 s of oracle time on the GPU box's host cores (two 12-frame 512^2 forwards + one ReferenceNet image).

@pytest.fixture(scope="module")
def cfg2_oracle(cfg2_models):
    """(inputs, oracle outputs) of one CFG-batched cfg2 UNet evaluation at t = 981, computed once for the f32 and the bf16 test:
    uncond row = plain forward under the uncond text; cond row = ReferenceNet write pass on the cond copy of the reference image
    (unet_controlnet.py:328-483 at F = 1, bank_mode='write') -> banks rounded through fp16 (mutual_self_attention.py:577,588) ->
    Backbone forward reading them under the cond text (mutual_self_attention.py:232-256)."""
    import time
    from oracle import unet_ref as U
    unet, ref = cfg2_models
    sd_u = {k: v.float().cpu() for k, v in unet._master.items()}
    sd_r = {k: v.float().cpu() for k, v in ref._master.items()}
    x = seeded_randn((1, 4, 12, 64, 64), 1)
    ctx = seeded_randn((2, 77, 768), 2)            # [uncond text, cond text]
    ref_lat = seeded_randn((1, 4, 64, 64), 3)
    t = 981
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    t0 = time.time()
    with torch.no_grad():
        y_uc = U.unet_forward(sd_u, cases.SD15_MOTION, x, t, ctx[:1])
        _, written = U.unet_forward(sd_r, cases.SD15, ref_lat[:, :, None], t, ctx[1:], bank_mode="write")
        banks = U.round_banks_fp16(written)
        order_r, order_u = U.transformer_block_order(cases.SD15), U.transformer_block_order(cases.SD15_MOTION)
        y_c = U.unet_forward(sd_u, cases.SD15_MOTION, x, t, ctx[1:], bank_mode="read", banks={pu: banks[pr] for pu, pr in zip(order_u, order_r)})
    print(f"oracle: uncond + ReferenceNet + cond forward at full size in {time.time() - t0:.0f} s")
    # BASELINE configs[2]: the same cond row with per-frame wav2vec audio tokens (12, 5, 768) as its attn2 context
    audio = seeded_randn((12, 5, 768), 4)
    with torch.no_grad():
        y_a = U.unet_forward(sd_u, cases.SD15_MOTION, x, t, ctx[1:], bank_mode="read", banks={pu: banks[pr] for pu, pr in zip(order_u, order_r)},
                             audio_features=audio)
    print(f"oracle: + the audio-context cond forward, {time.time() - t0:.0f} s in all")
    return dict(x=x, ctx=ctx, ref_lat=ref_lat, t=t, y=torch.cat([y_uc, y_c], 0), banks=[banks[p] for p in order_r], audio=audio, y_audio=y_a)


def _hip_cfg2_forward(unet, ref, o, audio=False):
    """The product's [uncond, cond] batch through the reference's own protocol: writer on [uncond, cond] copies of the reference
    image (EMOAnimationPipeline.py:711-716), reader.update(writer), one UNet call (EMOAnimationPipeline.py:759-790)."""
    from emote_hack_amd.reference_control import ReferenceAttentionControl
    writer = ReferenceAttentionControl(ref, do_classifier_free_guidance=True, mode="write", batch_size=1)
    reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", batch_size=1)
    try:
        ref(o["ref_lat"].repeat(2, 1, 1, 1).to(DEV), o["t"], encoder_hidden_states=o["ctx"].to(DEV), return_dict=False)
        banks = [writer.bank[p][0][1:].float().cpu() for p in writer.order]      # the cond copy's rows
        reader.update(writer)
        kw = {}
        if audio:   # [uncond row: a zero context, cond row: the audio tokens] per frame - what the pipeline binds for configs[2]
            kw["audio_features"] = torch.cat([torch.zeros_like(o["audio"]), o["audio"]]).to(DEV)
        y = unet(o["x"].repeat(2, 1, 1, 1, 1).to(DEV), o["t"], o["ctx"].to(DEV), _halves_identical=True, **kw).sample.float().cpu()
        reader.clear()
        writer.clear()
    finally:
        unet._reference_control = ref._reference_control = None
    return y, banks


def test_cfg2_full_unet_f32_vs_oracle(cfg2_models_f32, cfg2_oracle):
    """north_star's tolerance at the benchmarked size: the f32 mode of the SAME kernels (v_mfma_f32_32x32x2_f32) against the
    reference-pinned oracle at rtol 1e-3 / atol 1e-4 - every ReferenceNet bank, the uncond row and the bank-reading cond row."""
    u32, r32 = cfg2_models_f32
    y, banks = _hip_cfg2_forward(u32, r32, cfg2_oracle)
    for i, (b, b_ref) in enumerate(zip(banks, cfg2_oracle["banks"])):
        # the oracle's banks are already rounded through fp16, the writer's are rounded by update(): compare at fp16 resolution
        torch.testing.assert_close(b.half().float(), b_ref, rtol=2e-3, atol=2e-3, msg=lambda m, i=i: f"bank {i}: {m}")
    ref_y = cfg2_oracle["y"]
    e = (y - ref_y).abs()
    print(f"cfg2 full size f32 vs oracle: max {float(e.max()):.3e} mean {float(e.mean()):.3e} (mean |ref| {float(ref_y.abs().mean()):.3f})")
    torch.testing.assert_close(y, ref_y, rtol=1e-3, atol=1e-4)
    assert float((ref_y[0] - ref_y[1]).abs().mean()) > 1e-2       # the text and the banks are live in the oracle's cond row


def test_cfg2_full_unet_bf16_vs_oracle(cfg2_models, cfg2_oracle):
    """The BENCHMARKED dtype against the same oracle output: inside the low-precision yard-stick (the reference's own bf16 error on
    the tiny motion UNet relative to its output scale, tests/golden/unet_tiny.safetensors motion/out_bf16)."""
    unet, ref = cfg2_models
    y, _ = _hip_cfg2_forward(unet, ref, cfg2_oracle)
    ref_y = cfg2_oracle["y"]
    e = (y - ref_y).abs()
    print(f"cfg2 full size bf16 vs oracle: max {float(e.max()):.3e} mean {float(e.mean()):.3e} (mean |ref| {float(ref_y.abs().mean()):.3f}, "
          f"max |ref| {float(ref_y.abs().max()):.3f})")
    check(y, ref_y, torch.bfloat16)


def test_cfg3_full_unet_audio_context_f32_vs_oracle(cfg2_models_f32, cfg2_oracle):
    """BASELINE configs[2] at the benchmarked size: the bank-reading cond row with per-frame audio tokens as the attn2 context (f32 mode,
    rtol 1e-3 / atol 1e-4 against the oracle's forward with `audio_features`), and it differs from the text-context row."""
    u32, r32 = cfg2_models_f32
    y, _ = _hip_cfg2_forward(u32, r32, cfg2_oracle, audio=True)
    ref = cfg2_oracle["y_audio"]
    e = (y[1:] - ref).abs()
    print(f"cfg3 full size f32 vs oracle (cond row, audio context): max {float(e.max()):.3e} mean {float(e.mean()):.3e}")
    torch.testing.assert_close(y[1:], ref, rtol=1e-3, atol=1e-4)
    assert float((ref - cfg2_oracle["y"][1:]).abs().mean()) > 1e-3
