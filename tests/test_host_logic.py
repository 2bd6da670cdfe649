"""CPU tests of host-side logic that needs no device: the GEMM planner's split-K rules through the C ABI (pure host code in
libemo_hip.so) and the algebra of the two linear-map compositions the UNet packs at load time."""
import torch

from emote_hack_amd import _lib
from emote_hack_amd.synth import seeded_randn

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
