"""CPU tests of host-side logic that needs no device: the GEMM planner's split-K rules through the C ABI (pure host code in
libemo_hip.so) and the algebra of the two linear-map compositions the UNet packs at load time."""
import os

import pytest
import torch

from emote_hack_amd import _lib
from emote_hack_amd.synth import seeded_randn
from tests import cases

BF16, F32 = 1, 0


def _dt(name):
    from emote_hack_amd import ops
    return ops.dt(torch.zeros(1, dtype=name))


def test_planner_split_k_rules():
    """emo_gemm_suggest_split_k (csrc/gemm_api.h plan_gemm): the measured rules of DESIGN.md section 7."""
    lib = _lib.load()
    bf16, f32 = _dt(torch.bfloat16), _dt(torch.float32)
    sk = lambda M, N, K, dt=bf16, geglu=0, tr=0: lib.emo_gemm_suggest_split_k(M, N, K, dt, geglu, tr)
    # the 8x8-level 3x3 convs (M = 1536, K = 9 * 1280 / 9 * 2560): 30 tiles of 256x256, one block per CU -> 8 ways
    assert sk(1536, 1280, 11520) == 8
    assert sk(1536, 1280, 23040) == 8
    assert sk(640, 1280, 11520) == 17          # the ReferenceNet group (T = 10): 15 tiles -> 256 // 15 slices
    # f32 (validation mode) keeps the 128x128 tiles split to fill 512 slots
    assert sk(1536, 1280, 11520, f32) == 4
    # big single-pass shapes never split; short-K few-block shapes take 64x64 tiles unsplit (K = 2560 included)
    assert sk(98304, 2560, 320, geglu=1) == 1
    assert sk(1536, 1280, 1280) == 1
    assert sk(1536, 1280, 2560) == 1
    # long-K dense with few 128-row blocks still splits (ff.net.2 + proj_out tail of the 8x8 level: K = 6400)
    assert sk(1536, 1280, 6400) > 1
    # the workspace the caller must supply
    assert lib.emo_gemm_workspace_bytes(1536, 1280, 8) == 8 * 1536 * 1280 * 4
    assert lib.emo_gemm_workspace_bytes(1536, 1280, 1) == 0


def test_ff_tail_weights_compose_the_two_linears():
    """unet.ff_tail_weights: (g W2^T + b2 + h) Wo^T + bo == [g | h] Wt^T + bt, for a Linear and for a 1x1-conv proj_out."""
    from emote_hack_amd.unet import ff_tail_weights
    C, M = 24, 50
    g, h = seeded_randn((M, 4 * C), 1).double(), seeded_randn((M, C), 2).double()
    w2, b2 = seeded_randn((C, 4 * C), 3), seeded_randn((C,), 4)
    wo, bo = seeded_randn((C, C), 5), seeded_randn((C,), 6)
    ref = (g @ w2.double().t() + b2.double() + h) @ wo.double().t() + bo.double()
    for w_out in (wo, wo.reshape(C, C, 1, 1)):
        wt, bt = ff_tail_weights(w_out, bo, w2, b2)
        assert tuple(wt.shape) == (C, 5 * C) and tuple(bt.shape) == (C,) and wt.dtype == torch.float32
        got = torch.cat([g, h], 1) @ wt.double().t() + bt.double()
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def test_groupnorm_fold_algebra():
    """What emo_groupnorm_fold_linear builds (emo_hip.h): GN(x) W^T + b == x W'_n^T + b'_n per instance, with
    W'_n = W diag(gamma * rstd_n), b'_n = b + W beta - sum_c mean_n[c] W'_n[:, c] - restated in torch f64 against
    group_norm -> linear (the GPU test holds the kernel to the same statement)."""
    import torch.nn.functional as F
    N, S, C, G, Co = 3, 40, 32, 8, 12
    x = (seeded_randn((N, S, C), 7) * 1.7 + 0.4).double()
    gamma, beta = (1 + 0.1 * seeded_randn((C,), 8)).double(), (0.1 * seeded_randn((C,), 9)).double()
    w, b = seeded_randn((Co, C), 10).double(), seeded_randn((Co,), 11).double()
    ref = F.linear(F.group_norm(x.permute(0, 2, 1), G, gamma, beta, 1e-6).permute(0, 2, 1), w, b)
    xg = x.reshape(N, S, G, C // G)
    mean = xg.mean((1, 3))                                              # (N, G)
    rstd = (xg.var((1, 3), unbiased=False) + 1e-6).rsqrt()
    mean_c, rstd_c = mean.repeat_interleave(C // G, 1), rstd.repeat_interleave(C // G, 1)   # (N, C)
    wn = w[None] * (gamma[None] * rstd_c)[:, None, :]                   # (N, Co, C)
    bn = b[None] + (w @ beta)[None] - torch.einsum("nc,noc->no", mean_c, wn)
    got = torch.einsum("nsc,noc->nso", x, wn) + bn[:, None, :]
    torch.testing.assert_close(got, ref, rtol=1e-9, atol=1e-9)


def test_reference_group_size_never_exceeds_the_step_count():
    """ReferenceNet timesteps per batched pass (pipeline.reference_group_size): 1 <= T <= n_steps always; on several ranks a
    multiple of the world size whenever the step count allows it (10 over 8 ranks -> 16, never 16 of a 12-step loop), so no
    rank computes a padded timestep; and the groups tile the loop exactly."""
    from emote_hack_amd.pipeline import EMOAnimationPipeline as P
    for n_steps in (1, 2, 3, 7, 12, 20, 25, 50, 1000):
        for world in (1, 2, 3, 4, 8, 16):
            for want in (0, 1, 2, 5, 10, 25, 50, 10**6):
                T = P.reference_group_size(want, n_steps, world)
                assert 1 <= T <= n_steps, (want, n_steps, world, T)
                if world > 1 and T < n_steps:
                    assert T % world == 0, (want, n_steps, world, T)
                groups = [list(range(i, min(i + T, n_steps))) for i in range(0, n_steps, T)]
                assert sum(len(g) for g in groups) == n_steps and all(len(g) <= T for g in groups)
    assert P.reference_group_size(10, 50, 8) == 16 and P.reference_group_size(10, 12, 8) == 12 and P.reference_group_size(10, 50, 1) == 10


def test_strong_mode_deals_one_unit_per_rank_at_eight_gpus():
    """bench.py --mode strong = BASELINE configs[3]: one 48-frame clip = 4 windows x 2 CFG branches = 8 units; U[r::8] hands every
    rank exactly one, U[r::4] a [uncond, cond] pair of ONE window (shared prefix), U[r::2] two such pairs."""
    units = [(w, br) for br in (0, 1) for w in range(4)]
    assert all(len(units[r::8]) == 1 for r in range(8))
    for world in (4, 2, 1):
        for r in range(world):
            mine = units[r::world]
            byw = {}
            for w, br in mine:
                byw.setdefault(w, set()).add(br)
            assert all(v == {0, 1} for v in byw.values()), (world, r, mine)


def test_groupnorm_one_launch_plan():
    """emo_groupnorm_one_launch_ok (csrc/norm.hip gn1_geom): one workgroup per (instance, slab of whole groups) only where the slab
    fits <= 32 K elements, the launch has >= 32 workgroups and a slab's channels fill whole 16-byte vectors."""
    lib = _lib.load()
    bf16, f16, f32 = _dt(torch.bfloat16), _dt(torch.float16), _dt(torch.float32)
    ok = lib.emo_groupnorm_one_launch_ok
    # the bench's norms (N instances of S rows, C channels, 32 groups): 8x8 level joint, 16x16 per frame -> one launch
    assert ok(2, 12 * 64, 1280, 32, bf16) == 1 and ok(24, 256, 1280, 32, bf16) == 1 and ok(24, 64, 1280, 32, f16) == 1
    # larger instances keep the two coalesced passes: 16x16 joint, 32x32 per frame / joint, 64x64
    assert ok(2, 12 * 256, 1280, 32, bf16) == 0 and ok(24, 1024, 640, 32, bf16) == 0 and ok(2, 12 * 4096, 320, 32, bf16) == 0
    # too few workgroups (2 instances x 8 slabs of 4 groups at 10 channels per group)
    assert ok(2, 256, 320, 32, bf16) == 0 and ok(5, 37, 320, 32, bf16) == 1
    # f32: 4-wide vectors, slabs of 2 groups at 10 channels per group
    assert ok(5, 37, 320, 32, f32) == 1
    # outside the kernels' limits (groups, width, divisibility, dtype): never
    assert ok(4, 64, 1280, 256, bf16) == 0 and ok(4, 64, 1284, 32, bf16) == 0 and ok(4, 64, 1280, 32, 7) == 0 and ok(0, 64, 1280, 32, bf16) == 0


def test_geglu_polynomial_constants():
    """The erf-GELU polynomial of the bf16 GEGLU epilogue (csrc/common.h geglu_poly2), restated in numpy f32 with the constants read
    from the header: |gelu error| <= 1.9e-4 over the fitted range, x * P(x^2) >= 0.5 from |x| = 4 on (so the output clamp alone
    makes the tails exact - the kernel has no input clamp), monotone there, and no NaN up to overflow."""
    import os
    import re
    import math
    import numpy as np
    src = open(os.path.join(os.path.dirname(_lib.HERE), "emote_hack_amd", "csrc", "common.h")).read()
    body = src[src.index("void geglu_poly2("):]
    body = body[:body.index("oa = o.x")]
    coef = [float(m) for m in re.findall(r"v2f_t\{(-?[0-9.]+e[-+]?[0-9]+)f,", body)]
    lead = float(re.search(r"v2f_t p = \{([0-9.e+-]+)f,", body).group(1))
    coef = [lead] + coef
    assert len(coef) == 7, coef

    def E(t):
        t = t.astype(np.float32)
        u = t * t
        p = np.full_like(t, np.float32(coef[0]))
        for c in coef[1:]:
            p = p * u + np.float32(c)
        return t * p

    x = np.linspace(-6, 6, 240001, dtype=np.float32)
    g = x * (np.float32(0.5) + np.clip(E(x), -0.5, 0.5))
    ref = np.array([0.5 * v * (1.0 + math.erf(v / math.sqrt(2.0))) for v in x.astype(np.float64)])
    assert np.abs(g - ref).max() <= 1.9e-4
    t = np.logspace(math.log10(4.0), 19, 200001).astype(np.float32)
    with np.errstate(over="ignore"):
        e = E(t)
    assert not np.isnan(e).any() and e.min() >= 0.5 and (np.diff(e[np.isfinite(e)]) >= 0).all()


def test_context_windows_equal_the_oracle_over_a_sweep():
    """emote_hack_amd.context.uniform (an independent restatement: integer bit reversal, explicit start / hop walk) against
    oracle.scheduler_ref.uniform_windows (the reference's formulation, pinned by tests/golden/ints.json) over a sweep of geometries,
    `step` values (the pattern rotation) and both loop modes."""
    from emote_hack_amd.context import ordered_halving, uniform
    from oracle.scheduler_ref import ordered_halving as oh_ref, uniform_windows
    for v in (0, 1, 2, 3, 5, 8, 49, 1 << 40, (1 << 64) - 1):
        assert ordered_halving(v) == oh_ref(v)
    n_cases = 0
    for step in (0, 1, 2, 3, 7, 13):
        for n in (8, 12, 17, 24, 40, 48, 96):
            for size in (4, 12, 16):
                for stride in (1, 2, 3):
                    for ov in (0, 2, 4):
                        if ov >= size:
                            continue
                        for closed in (True, False):
                            got = list(uniform(step, 50, n, size, stride, ov, closed))
                            assert got == uniform_windows(step, 50, n, size, stride, ov, closed), (step, n, size, stride, ov, closed)
                            n_cases += 1
    assert n_cases > 1000


def test_pipeline_helper_methods_equal_the_reference_method_bodies():
    """The methods of EMOAnimationPipeline around __call__ (EMOAnimationPipeline.py:341-540) against goldens produced by running the
    reference's own method bodies on stub objects (tools/oracle/gen_golden.py gen_pipeline_methods): prepare_latents (host RNG, tiled
    clip noise), prepare_condition, next_step (DDIM inversion step), interpolate_latents with slerp / linear, select_controlnet_res_samples."""
    import numpy as np
    from safetensors.torch import load_file
    from emote_hack_amd import DDIMScheduler
    from emote_hack_amd import pipeline as P
    from emote_hack_amd.synth import seeded_randn
    g = load_file(os.path.join(cases.GOLDEN_DIR, "pipeline_methods.safetensors"))
    sch = DDIMScheduler()
    pipe = P.EMOAnimationPipeline(unet=type("U", (), {"device": torch.device("cpu")})(), scheduler=sch)
    sch.set_timesteps(50)
    x, eps = seeded_randn((4, 4, 16, 16), 500), seeded_randn((4, 4, 16, 16), 501)
    for t in (1, 21, 481, 981):
        xn, x0 = pipe.next_step(eps, t, x)
        torch.testing.assert_close(xn, g[f"next_step/{t}/x_next"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(x0, g[f"next_step/{t}/pred_x0"], rtol=1e-5, atol=1e-5)
    lat = pipe.prepare_latents(1, 4, 32, 64, 64, torch.float32, torch.device("cpu"), torch.Generator().manual_seed(5))
    assert torch.equal(lat, g["prepare_latents/out"])                      # the same host RNG stream, tiled the same way
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        pipe.prepare_latents(1, 4, 32, 64, 64, torch.float32, "cpu", None, latents=torch.zeros(1, 4, 32, 8, 8))
    cond = pipe.prepare_condition(g["prepare_condition/in"].numpy(), 1, "cpu", torch.float32, True)
    assert torch.equal(cond, g["prepare_condition/out"])
    l3 = seeded_randn((1, 4, 3, 4, 4), 502)
    assert pipe.interpolate_latents(l3, 1, "cpu") is l3                      # the factor __call__ hard-codes (:824)
    for name, is_slerp in (("slerp", True), ("linear", False)):
        P.set_tensor_interpolation_method(is_slerp)
        torch.testing.assert_close(pipe.interpolate_latents(l3, 3, "cpu"), g[f"interpolate/{name}"], rtol=1e-6, atol=1e-6)
    cache = {i: ([seeded_randn((1, 8, 4, 4), 600 + 10 * i + k) for k in range(3)], seeded_randn((1, 8, 2, 2), 700 + i)) for i in range(6)}
    down, mid = pipe.select_controlnet_res_samples(cache, [[0, 1], [4, 5]], True, 4, 2)
    assert all(torch.equal(d, g[f"select/down{k}"]) for k, d in enumerate(down)) and torch.equal(mid, g["select/mid"])
