"""GPU parity tests, kernel level: every C-ABI entry point vs a plain torch fp32 CPU statement of the
same op (the oracle's primitives), f32 mode at rtol 1e-3 / atol 1e-4 (north_star) and bf16 mode at a
measured, looser tolerance vs the SAME fp32 reference (bf16 has 8 mantissa bits; the reference's own
bf16 path misses 1e-3 by >10x - SURVEY.md section 7).  Calls go through emote_hack_amd.ops -> ctypes ->
libemo_hip.so."""
import math

import pytest
import torch
import torch.nn.functional as F

from emote_hack_amd.synth import seeded_randn

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: dict(rtol=1e-3, atol=1e-4), torch.bfloat16: dict(rtol=3e-2, atol=3e-2), torch.float16: dict(rtol=5e-3, atol=5e-3)}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def ops():
    from emote_hack_amd import ops as o
    return o


def q(t, dtype):
    """quantise a CPU fp32 tensor to the compute dtype's grid (inputs are identical on both sides)."""
    return t.to(dtype).float()


def close(got, ref, dtype, scale=1.0):
    tol = TOL[dtype]
    torch.testing.assert_close(got.float().cpu(), ref, rtol=tol["rtol"], atol=tol["atol"] * scale)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_roundtrip_and_concat(dtype):
    o = ops()
    x = q(seeded_randn((2, 5, 3, 6, 10), 1), dtype)
    rows = o.ncfhw_to_rows(x.to(DEV), dtype, cpad=8)
    ref = x.permute(0, 2, 3, 4, 1).reshape(-1, 5)
    close(rows[:, :5], ref, dtype)
    assert float(rows[:, 5:].float().abs().max()) == 0.0
    back = o.rows_to_ncfhw(rows[:, :5], 2, 5, 3, 6, 10)
    close(back, x, dtype)
    a, b = q(seeded_randn((37, 16), 2), dtype), q(seeded_randn((37, 24), 3), dtype)
    cat = o.concat_cols(a.to(DEV).to(dtype), b.to(DEV).to(dtype))
    close(cat, torch.cat([a, b], 1), dtype)
    s = o.add(a.to(DEV).to(dtype), a.to(DEV).to(dtype), alpha=0.5)
    close(s, a * 1.5, dtype)


def test_convert_fp16_round_and_silu():
    o = ops()
    x = seeded_randn((1000,), 4) * 3
    y = o.convert(x.to(DEV), torch.float32, fp16_round=True)
    assert torch.equal(y.cpu(), x.half().float())  # bit-exact IEEE half rounding
    yb = o.convert(x.to(DEV), torch.bfloat16)
    assert torch.equal(yb.cpu().float(), x.bfloat16().float())
    close(o.silu(x.to(DEV)), F.silu(x), torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_timestep_embedding(dtype):
    o = ops()
    from oracle.unet_ref import timestep_embedding
    ts = torch.tensor([981, 1, 500, 0, 999])
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    got = o.timestep_embedding(ts.to(DEV), freqs.to(DEV), 320, True, dtype)
    close(got, timestep_embedding(ts, 320, True, 0), dtype, scale=0.5 if dtype == torch.float32 else 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,S,C,G,silu", [(2, 4 * 64, 320, 32, True), (6, 100, 64, 8, False), (1, 12 * 16 * 16, 2560, 32, True),
                                          (3, 33, 1280, 32, True)])
def test_groupnorm(dtype, N, S, C, G, silu):
    o = ops()
    x = q(seeded_randn((N * S, C), 5) * 2 + 0.5, dtype)
    g, b = 1 + 0.1 * seeded_randn((C,), 6), 0.1 * seeded_randn((C,), 7)
    ref = F.group_norm(x.reshape(N, S, C).permute(0, 2, 1), G, g, b, 1e-5).permute(0, 2, 1).reshape(N * S, C)
    if silu:
        ref = F.silu(ref)
    got = o.group_norm(x.to(DEV).to(dtype), g.to(DEV), b.to(DEV), N, G, 1e-5, silu)
    close(got, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,S,C,G,silu", [(2, 12 * 8 * 8, 1280, 32, True), (24, 16 * 16, 1280, 32, False), (24, 8 * 8, 1280, 32, True),
                                          (2, 12 * 8 * 8, 640, 32, True), (5, 37, 320, 32, True), (4, 250, 960, 32, False),
                                          (3, 400, 2560, 32, True), (40, 1000, 64, 8, False)])
def test_groupnorm_one_launch(dtype, N, S, C, G, silu):
    """emo_groupnorm (one workgroup per (instance, slab of whole groups), the tensor held in registers between the statistics
    and the normalisation) at the bench's 8x8 / per-frame 16x16 geometries, ragged row counts and slabs of 1 / 2 / 4 groups:
    against F.group_norm in f32 and against the two-launch path on the same input; also into a strided output."""
    from emote_hack_amd import _lib
    o = ops()
    lib = _lib.load()
    dti = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    assert lib.emo_groupnorm_one_launch_ok(N, S, C, G, dti) == 1
    x = q(seeded_randn((N * S, C), 5) * (1 + torch.arange(C) % 7 * 0.2) + 0.5, dtype)
    g, b = 1 + 0.1 * seeded_randn((C,), 6), 0.1 * seeded_randn((C,), 7)
    ref = F.group_norm(x.reshape(N, S, C).permute(0, 2, 1), G, g, b, 1e-5).permute(0, 2, 1).reshape(N * S, C)
    if silu:
        ref = F.silu(ref)
    xd, gd, bd = x.to(DEV).to(dtype), g.to(DEV), b.to(DEV)
    got = o.group_norm(xd, gd, bd, N, G, 1e-5, silu)
    close(got, ref, dtype)
    # the two-launch path on the same rows
    part = torch.empty(lib.emo_groupnorm_workspace_bytes(N, S, C, G) // 4, device=DEV, dtype=torch.float32)
    two = torch.empty_like(xd)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.emo_groupnorm_stats(xd.data_ptr(), C, part.data_ptr(), N, S, C, G, dti, st) == 0
    assert lib.emo_groupnorm_apply(xd.data_ptr(), C, part.data_ptr(), gd.data_ptr(), bd.data_ptr(), two.data_ptr(), C, N, S, C, G, 1e-5,
                                   int(silu), dti, st) == 0
    close(got, two.float().cpu(), dtype)
    # strided output (the concat buffers of the up path)
    wide = torch.zeros(N * S, C + 64, device=DEV, dtype=dtype)
    o.group_norm(xd, gd, bd, N, G, 1e-5, silu, out=wide[:, 32:32 + C])
    assert torch.equal(wide[:, 32:32 + C], got) and not wide[:, :32].any() and not wide[:, 32 + C:].any()
    # a geometry that does not fit is refused by the one-launch entry, not mangled
    assert lib.emo_groupnorm_one_launch_ok(2, 12 * 64 * 64, 320, 32, dti) == 0
    assert lib.emo_groupnorm(xd.data_ptr(), 320, gd.data_ptr(), bd.data_ptr(), two.data_ptr(), 320, 2, 12 * 64 * 64, 320, 32, 1e-5, 0, dti, st) != 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,S,C,G,Cout", [(6, 1024, 320, 32, 320), (3, 256, 64, 8, 96), (2, 4096, 640, 32, 640), (5, 256, 1280, 32, 128)])
def test_groupnorm_folded_into_linear(dtype, N, S, C, G, Cout):
    """emo_groupnorm_fold_linear + emo_gemm(w_slab_rows): per-frame GroupNorm (eps 1e-6, no activation) followed by proj_in
    (attention.py:124,135-146) as per-instance weights over the RAW rows - against group_norm -> linear in f32.  The inputs have
    a mean of the size of their spread (the mean term rides in the per-instance bias)."""
    o = ops()
    x = q(seeded_randn((N * S, C), 5) * (1 + torch.arange(N).repeat_interleave(S)[:, None] * 0.3) + 0.7, dtype)
    g, b = 1 + 0.1 * seeded_randn((C,), 6), 0.1 * seeded_randn((C,), 7)
    w, bias = q(seeded_randn((Cout, C), 8) / C ** 0.5, dtype), 0.1 * seeded_randn((Cout,), 9)
    ref = F.linear(F.group_norm(x.reshape(N, S, C).permute(0, 2, 1), G, g, b, 1e-6).permute(0, 2, 1).reshape(N * S, C), w, bias)
    xd = x.to(DEV).to(dtype)
    wn, rb = o.group_norm_fold_linear(xd, g.to(DEV), b.to(DEV), N, G, 1e-6, w.to(DEV).to(dtype), bias.to(DEV))
    assert tuple(wn.shape) == (N, Cout, C) and tuple(rb.shape) == (N, Cout)
    got = o.gemm(xd, wn, rb, w_slab_rows=S)
    close(got, ref, dtype, scale=2.0)
    # the unfused pair is the same function
    close(o.gemm(o.group_norm(xd, g.to(DEV), b.to(DEV), N, G, 1e-6, False), w.to(DEV).to(dtype), bias.to(DEV)), ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nb,L,C,tile", [(24, 4096, 320, None), (6, 1024, 640, None), (3, 256, 1280, None), (2, 256, 64, None), (2, 512, 320, 2), (4, 256, 96, 1), (2, 256, 320, 3)])
def test_gemm_qkv_one_launch_with_transposed_v(dtype, nb, L, C, tile):
    """emo_gemm_params.vt: the LayerNorm-folded q | k | v projection of a self-attention (orig_attention.py:598-600) as ONE launch - the
    columns [0, 2C) row-major, the V columns [2C, 3C) stored transposed per batch of L rows by the V waves of the same tiles - against the
    two launches it replaces (q | k row-major, V^T through the transposed-store kernel): the same accumulators, hence the same bits, and
    against LayerNorm -> linear in f32.  Shapes of the path (256x256 phase-loop tiles at the three big levels), a small one, and pinned
    128x128 / 64x64 / 128x160 tiles; a split the planned tile does not serve is refused (None), not mangled."""
    o = ops()
    if dtype == torch.float32 and nb * L > 30000:
        nb = 4
    M = nb * L
    x = q(seeded_randn((M, C), 5) * (1 + torch.arange(C) % 5 * 0.25) + 0.3, dtype)
    g, b = 1 + 0.1 * seeded_randn((C,), 6), 0.1 * seeded_randn((C,), 7)
    w = seeded_randn((3 * C, C), 8) / C ** 0.5
    ref = F.linear(F.layer_norm(x, (C,), g, b, 1e-5), w)
    xd = x.to(DEV).to(dtype)
    wp, cs, bp = (t.to(DEV).contiguous() for t in _ln_fold(w, None, g, b, dtype))
    st = o.layer_norm_stats(xd, 1e-5)
    ld = (L + 7) // 8 * 8
    both = o.gemm(xd, wp, bp, ln=(cs, st), vt_cols=C, vt_rows=L, vt_ld=ld, tile=tile)
    if tile == 3:       # 128x160 tiles: a wave covers 160 columns - 2C = 640 = 4 x 160 is served, checked below with C = 320
        assert both is not None
    assert both is not None
    qk, vt = both
    assert tuple(qk.shape) == (M, 2 * C) and tuple(vt.shape) == (nb, C, ld)
    qk2 = o.gemm(xd, wp[:2 * C].contiguous(), bp[:2 * C].contiguous(), ln=(cs[:2 * C].contiguous(), st), tile=tile)
    vt2 = o.gemm(xd, wp[2 * C:].contiguous(), bp[2 * C:].contiguous(), ln=(cs[2 * C:].contiguous(), st), transpose_rows=L, transpose_ld=ld)
    close(qk, ref[:, :2 * C], dtype, scale=2.0)
    close(vt[:, :, :L].transpose(1, 2).reshape(M, C), ref[:, 2 * C:], dtype, scale=2.0)
    assert torch.equal(qk, qk2)
    # (the transposed-store kernel multiplies with the operands swapped - W rows as the MFMA's B operand; bf16 and f32 give the same bits either
    # way, the f16 MFMA now and then an accumulator that rounds to the neighbouring half)
    d = (vt[:, :, :L].float() - vt2[:, :, :L].float()).abs()
    assert float((d > 0).float().mean()) < 1e-3 and float(d.max()) <= 2 * torch.finfo(dtype).eps * float(vt2[:, :, :L].float().abs().max())
    if dtype != torch.float16:
        assert torch.equal(vt[:, :, :L], vt2[:, :, :L])
    # a V block that does not start on a wave boundary of the planned tile
    assert o.gemm(xd, wp, bp, ln=(cs, st), vt_cols=C - 8, vt_rows=L, vt_ld=ld, tile=tile) is None


@pytest.mark.parametrize("tile", [None, 7, 4])
@pytest.mark.parametrize("N,S,C,Cout", [(14, 4096, 64, 256), (8, 4096, 128, 512)])
def test_weight_slabs_on_the_ping_pong_tile(N, S, C, Cout, tile):
    """Per-instance weight slabs on the 256x256 tiles the planner picks for BIG outputs (N % 256 == 0, >= 224 tiles: widths 256 / 512
    at 64x64 latents) - the ping-pong loader formed its W addresses from the base pointer, i.e. every frame was multiplied by
    frame 0's folded weights (and got its own bias).  Also with the tile pinned (7 = ping-pong, 4 = lockstep 256x256).  bf16
    against per-slab f32 linears; the slabs differ by construction."""
    o = ops()
    dtype = torch.bfloat16
    x = q(seeded_randn((N * S, C), 5), dtype)
    w = q(seeded_randn((N, Cout, C), 8) / C ** 0.5 * (1 + torch.arange(N)[:, None, None] * 0.25), dtype)
    bias = 0.1 * seeded_randn((N, Cout), 9)
    ref = torch.cat([F.linear(x[i * S:(i + 1) * S], w[i], bias[i]) for i in range(N)])
    got = o.gemm(x.to(DEV).to(dtype), w.to(DEV).to(dtype), bias.to(DEV), w_slab_rows=S, tile=tile)
    close(got, ref, dtype, scale=2.0)


def test_gemm_weight_slab_geometry_is_checked():
    from emote_hack_amd._lib import EmoHipError
    o = ops()
    a = torch.zeros(512, 64, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(2, 64, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(EmoHipError):
        o.gemm(a[:384], w[:1].expand(3, 64, 64).contiguous(), None, w_slab_rows=128)     # slabs must be multiples of 256 rows
    o.gemm(a, w, None, w_slab_rows=256)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,pe", [(100, 320, False), (2 * 3 * 16, 64, True), (77, 1280, False), (12 * 4, 640, True)])
def test_layernorm(dtype, M, C, pe):
    o = ops()
    x = q(seeded_randn((M, C), 8) * 2 + 0.3, dtype)
    g, b = 1 + 0.1 * seeded_randn((C,), 9), 0.1 * seeded_randn((C,), 10)
    ref = F.layer_norm(x, (C,), g, b)
    pe_t = None
    if pe:
        frames, rpf = 3, M // 6   # rows = 2 batches x 3 frames x rpf pixels
        pe_t = seeded_randn((24, C), 11)
        fr = (torch.arange(M) // rpf) % frames
        ref = q(ref, dtype) + pe_t[fr]
    got = o.layer_norm(x.to(DEV).to(dtype), g.to(DEV), b.to(DEV), pe=pe_t.to(DEV) if pe else None,
                       rows_per_frame=rpf if pe else 0, frames=frames if pe else 0)
    close(got, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (128, 128, 64), (77, 1280, 768), (2, 1280, 320), (1000, 96, 2560)])
def test_gemm_bias_residual(dtype, M, N, K):
    o = ops()
    a, w = q(seeded_randn((M, K), 12), dtype), q(seeded_randn((N, K), 13) / math.sqrt(K), dtype)
    bias, res = 0.1 * seeded_randn((N,), 14), q(seeded_randn((M, N), 15), dtype)
    ref = (F.linear(a, w, bias) + res) * 0.5
    got = o.gemm(a.to(DEV).to(dtype), w.to(DEV).to(dtype), bias.to(DEV), residual=res.to(DEV).to(dtype), out_scale=0.5)
    close(got, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,geglu,res", [(8192 + 37, 2560, 320, True, False),     # 256x256 tiles, ragged last row panel, GEGLU
                                             (70000, 640, 192, False, True),          # > 512 tiles: several tiles per persistent block
                                             (40000, 320, 320, False, True),          # 128x160 tiles
                                             (33000, 1280, 128, False, False)])       # 256x256, plain copy epilogue
def test_gemm_persistent_many_tiles(dtype, M, N, K, geglu, res):
    """The persistent loop (a block walks several tiles, the loader streams across tile boundaries), the LDS-staged
    coalesced epilogue and the bias-initialised accumulators, against a torch f32 matmul on the same device."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(123)
    a = q(torch.randn(M, K, generator=g), dtype).to(DEV)
    w = q(torch.randn(N, K, generator=g) / math.sqrt(K), dtype).to(DEV)
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV)
    n_out = N // 2 if geglu else N
    r = q(torch.randn(M, n_out, generator=g), dtype).to(DEV) if res else None
    y = a.float() @ w.float().t() + bias
    if geglu:   # weight rows interleaved (32 value, 32 gate): undo on the reference side
        y = y.reshape(M, N // 64, 2, 32)
        y = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(M, n_out)
    if res:
        y = y + r.float()
    got = o.gemm(a.to(dtype), w.to(dtype), bias, geglu=geglu, residual=r.to(dtype) if res else None)
    close(got, y.cpu(), dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,geglu,res,ln", [(8192 + 37, 2560, 320, True, False, True),   # GEGLU + LayerNorm fold, ragged last row panel
                                                (140000, 512, 192, False, True, False),     # 1100 tiles: ~4 per persistent block, 3 stages each
                                                (33000, 1280, 64, False, False, False),     # ONE K-tile per tile (the phase loader needs two: lockstep loop)
                                                (33000, 1280, 128, False, False, False),    # TWO K-tiles per tile: every request crosses a tile boundary
                                                (5000, 960, 1600, False, True, False),      # ragged last column panel, 25 stages
                                                (6144, 3840, 1280, False, False, True)])
def test_gemm_ping_pong_main_loop(dtype, M, N, K, geglu, res, ln):
    """EMO_TILE_256x256_PP: the PHASE main loop (gemm_impl.h EMO_GEMM_PH) - four quadrant phases per K-tile, each a load section
    (fragment reads + the LDS-DMA requests of one quarter of a later K-tile) and a pure-MFMA section, the two wave rows of a block
    one barrier apart, quarters restaged behind their last read, the K-tile stream running on into the block's next tile.  Every
    ordering rule of that loop is a data hazard when broken - checked on many tiles per block, two- and three-K-tile tiles, ragged
    edges, against the f32 matmul and BIT FOR BIT against the lockstep loop of the same tile (same MFMA order per accumulator)."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(321)
    a = q(torch.randn(M, K, generator=g) * 1.5 + 0.2, dtype).to(DEV).to(dtype)
    wf = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = 0.1 * torch.randn(N, generator=g)
    kw = {}
    if ln:
        gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
        wp, cs, bp = _ln_fold(wf, bias, gamma, beta, dtype)
        y = F.linear(F.layer_norm(a.float().cpu(), (K,), gamma, beta), wf.to(dtype).float(), bias)
        w, bias_d = wp.to(DEV).contiguous(), bp.to(DEV)
        kw["ln"] = (cs.to(DEV).contiguous(), o.layer_norm_stats(a))
    else:
        w, bias_d = wf.to(dtype).to(DEV), bias.to(DEV)
        y = a.float().cpu() @ w.float().cpu().t() + bias
    n_out = N // 2 if geglu else N
    r = q(torch.randn(M, n_out, generator=g), dtype).to(DEV).to(dtype) if res else None
    if geglu:
        y = y.reshape(M, N // 64, 2, 32)
        y = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(M, n_out)
    if res:
        y = y + r.float().cpu()
    for rep in range(3):     # a race shows on some launches only
        got = o.gemm(a, w, bias_d, geglu=geglu, residual=r, tile=7, split_k=1, **kw)
        lock = o.gemm(a, w, bias_d, geglu=geglu, residual=r, tile=4, split_k=1, **kw)
        assert torch.equal(got, lock), f"rep {rep}: {int((got != lock).sum())} elements differ from the lockstep loop"
    close(got, y, dtype, scale=2.0 if ln else 1.0)


@pytest.mark.parametrize("M,N,K,tile", [(128 * 13, 1920, 192, 2), (128 * 13 + 5, 2240, 128, 3), (256 * 5 + 37, 3584, 128, 4),
                                        (256 * 9, 4096, 64, 0), (64 * 21 + 3, 1024, 64, 1)])
def test_gemm_grouped_tile_order_covers_every_tile(M, N, K, tile):
    """Wide outputs (more than 12 column tiles) walk the tiles in bands of 4 / 8 tile rows, m fastest inside a band
    (gemm_impl.h tile_mn); the last band is short when the tile rows do not divide - every output element must still be written
    exactly once (a poisoned output buffer would show through), for every tile shape, and with the order pinned both ways."""
    o = ops()
    dtype = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(7)
    a = q(torch.randn(M, K, generator=g), dtype).to(DEV)
    w = q(torch.randn(N, K, generator=g) / math.sqrt(K), dtype).to(DEV)
    y = (a.float() @ w.float().t()).cpu()
    for code in (tile, tile + 16 * 1, tile + 16 * 4, tile + 16 * 8):   # planned order, n fastest, bands of 4, bands of 8
        out = torch.full((M, N), float("nan"), device=DEV, dtype=dtype)
        o.gemm(a.to(dtype), w.to(dtype), None, out=out, tile=code, split_k=1)
        assert bool(torch.isfinite(out).all()), f"tile code {code}: unwritten outputs"
        close(out, y, dtype)


@pytest.mark.parametrize("M,N,tile", [(98304, 320, 3), (98304, 320, 2), (3072, 5120, 2), (24576, 640, 0), (6144, 1280, 4)])
def test_gemm_residual_passes_through_exactly_when_the_product_is_zero(M, N, tile):
    """A = 0, no bias: the output must equal the residual BIT FOR BIT, on every element, with several tiles per persistent
    block.  Regression for the staged epilogue's store path: a 16-byte buffer store whose data registers were overwritten by
    the next chunk's unpack wrote `x << 16` patterns into the last tile of a pass for some lanes (gemm_impl.h epilogue_lds) -
    far inside the bf16 tolerance of a value comparison for small |x|, so compared exactly here."""
    o = ops()
    dtype = torch.bfloat16
    a = torch.zeros(M, 64, device=DEV, dtype=dtype)
    w = torch.ones(N, 64, device=DEV, dtype=dtype)
    r = (torch.arange(M, device=DEV)[:, None] % 251 + (torch.arange(N, device=DEV)[None, :] // 8) * 0.5).to(dtype)
    for _ in range(2):
        out = torch.full((M, N), 777.0, device=DEV, dtype=dtype)
        o.gemm(a, w, None, residual=r, out=out, tile=tile, split_k=1)
        assert torch.equal(out, r)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric_identity(dtype):
    """A = I with an asymmetric W catches a transposed C-write (guide: always A=I-check with asymmetric B)."""
    o = ops()
    K = 128
    a = torch.eye(K)
    w = q(seeded_randn((192, K), 16), dtype)
    got = o.gemm(a.to(DEV).to(dtype), w.to(DEV).to(dtype))
    close(got, w.t().contiguous(), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_rowbias_geglu_transpose(dtype):
    o = ops()
    M, K, No = 2 * 96, 64, 128
    a = q(seeded_randn((M, K), 17), dtype)
    w, bias = q(seeded_randn((2 * No, K), 18) / 8, dtype), 0.1 * seeded_randn((2 * No,), 19)
    hv, gate = F.linear(a, w, bias).chunk(2, dim=-1)
    ref = hv * F.gelu(gate)
    wi = torch.cat([w[:No].reshape(No // 32, 32, K), w[No:].reshape(No // 32, 32, K)], 1).reshape(2 * No, K)
    bi = torch.cat([bias[:No].reshape(No // 32, 32), bias[No:].reshape(No // 32, 32)], 1).reshape(2 * No)
    got = o.gemm(a.to(DEV).to(dtype), wi.to(DEV).to(dtype).contiguous(), bi.to(DEV).contiguous(), geglu=True)
    close(got, ref, dtype)
    # row bias (temb) per batch of 96 rows
    rb = seeded_randn((2, 2 * No), 20)
    ref2 = F.linear(a, w) + rb.repeat_interleave(96, 0)
    got2 = o.gemm(a.to(DEV).to(dtype), w.to(DEV).to(dtype), None, rowbias=rb.to(DEV), rows_per_batch=96)
    close(got2, ref2, dtype)
    # V^T store: (M/L, N, ld)
    L = 48
    got3 = o.gemm(a.to(DEV).to(dtype), w.to(DEV).to(dtype), transpose_rows=L, transpose_ld=48)
    ref3 = F.linear(a, w).reshape(M // L, L, 2 * No).permute(0, 2, 1)
    close(got3[:, :, :L], ref3, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 2, 4])
def test_geglu_gate_range_and_tails(dtype, tile):
    """The GEGLU epilogue's erf-GELU (common.h geglu_poly2: gelu(x) = x * (0.5 + clamp(x * P(x^2), -0.5, 0.5)), no input clamp)
    over the whole gate range: gates on a grid through [-12, 12] (the fitted range is |x| <= 4), at +-50 / +-3e4 / +-1e30 and 0,
    values +-1 and 3 - against value * gelu(gate) in f64; beyond |gate| = 9 exactly value * gate / 0."""
    o = ops()
    M, K = 512, 64
    big = 1e30 if dtype == torch.bfloat16 else 6e4       # (x^2 overflows f32 at 1e30: the clamp still decides)
    gates = torch.cat([torch.linspace(-12, 12, M - 8), torch.tensor([0.0, -0.0, 50.0, -50.0, 3e4, -3e4, big, -big])])
    vals = torch.tensor([1.0, -1.0, 3.0])[torch.arange(M) % 3]
    a = torch.zeros(M, K)
    a[:, 0], a[:, 1] = gates, vals
    a = q(a, dtype)
    w = torch.zeros(64, K)          # rows 0-31: value rows (pick column 1), rows 32-63: gate rows (pick column 0)
    w[:32, 1], w[32:, 0] = 1.0, 1.0
    got = o.gemm(a.to(DEV).to(dtype), w.to(DEV).to(dtype), None, geglu=True, tile=tile).float().cpu()
    g, v = a[:, 0].double(), a[:, 1].double()
    ref = (v * 0.5 * g * (1.0 + torch.erf(g / math.sqrt(2.0)))).float()
    ref = ref.to(dtype).float()
    assert tuple(got.shape) == (M, 32) and torch.equal(got, got[:, :1].expand(M, 32))
    y = got[:, 0]
    fin = ref.isfinite()
    assert torch.equal(y.isfinite(), fin) and torch.equal(y[~fin], ref[~fin])          # value * (+1e30) overflows the 2-byte type alike
    tol = TOL[dtype]
    torch.testing.assert_close(y[fin], ref[fin], rtol=tol["rtol"], atol=2e-3)
    far = fin & (g.abs() > 9)        # (erf(9 / sqrt 2) rounds to 1 in f64: the reference is value * gate / 0 exactly there)
    assert torch.equal(y[far], ref[far])                                                 # the tails are exact: value * gate and 0
    assert torch.equal(y[g == 0], torch.zeros(int((g == 0).sum())))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,H,W,Cin,Cout,stride,up", [(3, 8, 8, 32, 64, 1, False), (2, 16, 12, 64, 32, 2, False),
                                                      (2, 4, 6, 64, 64, 1, True), (1, 16, 16, 8, 320, 1, False),
                                                      (2, 8, 8, 320, 4, 1, False), (1, 6, 6, 960, 320, 1, False)])
def test_conv3x3(dtype, n, H, W, Cin, Cout, stride, up):
    o = ops()
    x = q(seeded_randn((n, Cin, H, W), 21), dtype)
    wt, bias = q(seeded_randn((Cout, Cin, 3, 3), 22) / math.sqrt(9 * Cin), dtype), 0.1 * seeded_randn((Cout,), 23)
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xi, wt, bias, stride=stride, padding=1)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    got, Ho, Wo = o.conv3x3(rows.to(DEV).to(dtype), wp.to(DEV).to(dtype).contiguous(), bias.to(DEV), n, H, W, stride=stride,
                            upsample2x=up)
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    close(got, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,H,W,Cin,Cout", [(2, 8, 16, 64, 128), (3, 16, 32, 128, 320), (1, 24, 16, 192, 64), (13, 16, 16, 64, 132),
                                            (3, 24, 24, 128, 320), (2, 12, 24, 64, 128), (2, 20, 40, 64, 192), (1, 9, 17, 64, 64)])
def test_conv3x3_halo_path(dtype, n, H, W, Cin, Cout):
    """Stride-1 3x3 with Cin a multiple of the 128-byte channel chunk: the halo-reuse kernel (csrc/conv_halo_impl.h), with the resnet
    epilogue (bias + temb row bias + residual; resnet.py:188,197).  Frames of whole 8 x 16 patches, and - the 24 x 24 level of
    BASELINE configs[4], odd sizes - frames whose LAST patch row / column is shifted back inside the frame (two blocks store the
    overlapped pixels, bit-identically)."""
    o = ops()
    x = q(seeded_randn((n, Cin, H, W), 31), dtype)
    wt, bias = q(seeded_randn((Cout, Cin, 3, 3), 32) / math.sqrt(9 * Cin), dtype), 0.1 * seeded_randn((Cout,), 33)
    res = q(seeded_randn((n, Cout, H, W), 34), dtype)
    rb = seeded_randn((n, Cout), 35)
    ref = F.conv2d(x, wt, bias, padding=1) + rb[:, :, None, None] + res
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    rrows = res.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous()
    got, Ho, Wo = o.conv3x3(rows.to(DEV).to(dtype), wp.to(DEV).to(dtype).contiguous(), bias.to(DEV), n, H, W,
                            rowbias=rb.to(DEV), rows_per_batch=H * W, residual=rrows.to(DEV).to(dtype), split_k=1)
    assert (Ho, Wo) == (H, W)
    close(got, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ph", [8, 16])
@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 16, 32, 128, 320), (2, 32, 16, 64, 132), (5, 16, 16, 192, 128)])
def test_conv3x3_halo_patch_heights(dtype, ph, n, H, W, Cin, Cout):
    """Both patch heights of the halo kernel (8 rows / 4 waves and 16 rows / 8 waves; the planner picks by tile count, the test
    pins them through emo_gemm_params.tile) on the same inputs - several patches per frame, ragged N, temb row bias per frame."""
    o = ops()
    x = q(seeded_randn((n, Cin, H, W), 131), dtype)
    wt, bias = q(seeded_randn((Cout, Cin, 3, 3), 132) / math.sqrt(9 * Cin), dtype), 0.1 * seeded_randn((Cout,), 133)
    res = q(seeded_randn((n, Cout, H, W), 134), dtype)
    rb = seeded_randn((n, Cout), 135)
    ref = F.conv2d(x, wt, bias, padding=1) + rb[:, :, None, None] + res
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    rrows = res.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous()
    got, _, _ = o.conv3x3(rows.to(DEV).to(dtype), wp.to(DEV).to(dtype).contiguous(), bias.to(DEV), n, H, W,
                          rowbias=rb.to(DEV), rows_per_batch=H * W, residual=rrows.to(DEV).to(dtype), split_k=1, tile=1 if ph == 8 else 2)
    close(got, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype)
    if ph == 16 and Cout % 128 == 64:
        # widths that are odd multiples of 64 with 16-row patches: the last 192 columns on 192-channel blocks (planned when one block per
        # patch is a single round - these shapes - or forced by tile bit 4) vs the 64-column remainder launch (tile bit 3): the same conv
        got_64, _, _ = o.conv3x3(rows.to(DEV).to(dtype), wp.to(DEV).to(dtype).contiguous(), bias.to(DEV), n, H, W,
                                 rowbias=rb.to(DEV), rows_per_batch=H * W, residual=rrows.to(DEV).to(dtype), split_k=1, tile=2 | 8)
        got_192, _, _ = o.conv3x3(rows.to(DEV).to(dtype), wp.to(DEV).to(dtype).contiguous(), bias.to(DEV), n, H, W,
                                  rowbias=rb.to(DEV), rows_per_batch=H * W, residual=rrows.to(DEV).to(dtype), split_k=1, tile=2 | 16)
        assert torch.equal(got_64, got_192) and torch.equal(got, got_192)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ph", [8, 16])
@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 16, 32, 128, 320), (2, 16, 16, 64, 192), (2, 16, 16, 64, 64), (4, 16, 16, 128, 576)])
def test_conv3x3_halo_remainder_launch_is_the_same_conv(dtype, ph, n, H, W, Cin, Cout):
    """A width that is an odd multiple of 64 (320 = 128 + 128 + 64) runs as two launches: the 128-column tiles and a launch of
    64-channel blocks for the remainder columns (emo_gemm_params.tile bit 2 keeps the single launch whose last tile multiplies 64
    columns of zeros).  Same arithmetic per output element: bit-identical to the single launch, and equal to the f32 conv with
    bias + temb row bias + residual; N = 64 alone takes the 64-channel blocks directly."""
    o = ops()
    x = q(seeded_randn((n, Cin, H, W), 231), dtype)
    wt, bias = q(seeded_randn((Cout, Cin, 3, 3), 232) / math.sqrt(9 * Cin), dtype), 0.1 * seeded_randn((Cout,), 233)
    res = q(seeded_randn((n, Cout, H, W), 234), dtype)
    rb = seeded_randn((n, Cout), 235)
    ref = F.conv2d(x, wt, bias, padding=1) + rb[:, :, None, None] + res
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV).to(dtype)
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV).to(dtype).contiguous()
    rrows = res.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV).to(dtype)
    pin = 1 if ph == 8 else 2
    kw = dict(rowbias=rb.to(DEV), rows_per_batch=H * W, residual=rrows, split_k=1)
    got, _, _ = o.conv3x3(rows, wp, bias.to(DEV), n, H, W, tile=pin, **kw)
    one, _, _ = o.conv3x3(rows, wp, bias.to(DEV), n, H, W, tile=pin | 4, **kw)
    close(got, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype)
    assert torch.equal(got, one)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ph", [8, 16])
@pytest.mark.parametrize("B,Fr,H,W,Cin,Cout,lda_pad,silu", [(2, 3, 16, 32, 128, 320, 0, True), (1, 2, 16, 16, 64, 64, 64, True), (3, 1, 32, 16, 192, 132, 0, True),
                                                           (2, 2, 16, 16, 320, 128, 320, False), (1, 13, 16, 16, 64, 192, 0, True)])
def test_conv3x3_groupnorm_silu_inside_the_conv(dtype, ph, B, Fr, H, W, Cin, Cout, lda_pad, silu):
    """resnet.py:180-183,191-196 norm -> nonlinearity -> conv with the normalisation INSIDE the halo-reuse conv (emo_gemm_params.gn_coef):
    the conv reads the RAW rows, each halo chunk is normalised, activated and rounded in LDS behind its direct-to-LDS load - against
    F.group_norm (statistics joint over the Fr frames of a batch row) -> silu -> conv2d in f32, and against the unfused pair of
    launches on the same input, which it reproduces up to a rounding boundary here and there in the 2-byte modes (same factors, same
    arithmetic, same rounding of the normalised tensor; the padding stays zero).  Both patch heights, the 64-column remainder launch (N = 320, 132, 192), an input
    that is the left part of a wider buffer (the concat buffers of the up path), several instances, temb row bias + residual."""
    o = ops()
    n, G = B * Fr, 32
    x = q(seeded_randn((n, Cin, H, W), 331) * (1 + torch.arange(Cin)[None, :, None, None] % 5 * 0.3) + 0.4, dtype)
    g, b = 1 + 0.1 * seeded_randn((Cin,), 336), 0.1 * seeded_randn((Cin,), 337)
    wt, bias = q(seeded_randn((Cout, Cin, 3, 3), 332) / math.sqrt(9 * Cin), dtype), 0.1 * seeded_randn((Cout,), 333)
    res = q(seeded_randn((n, Cout, H, W), 334), dtype)
    rb = seeded_randn((B, Cout), 335)
    xn = F.group_norm(x.reshape(B, Fr, Cin, H, W).permute(0, 2, 1, 3, 4), G, g, b, 1e-5).permute(0, 2, 1, 3, 4).reshape(n, Cin, H, W)
    if silu:
        xn = F.silu(xn)
    ref = F.conv2d(xn, wt, bias, padding=1) + rb.repeat_interleave(Fr, 0)[:, :, None, None] + res
    wide = torch.zeros(n * H * W, Cin + lda_pad, device=DEV, dtype=dtype)
    wide[:, :Cin] = x.permute(0, 2, 3, 1).reshape(-1, Cin).to(DEV).to(dtype)
    if lda_pad:
        wide[:, Cin:] = 7.0          # the neighbour's columns must not leak into the statistics or the halo
    rows = wide[:, :Cin]
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV).to(dtype).contiguous()
    rrows = res.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV).to(dtype)
    gd, bd = g.to(DEV), b.to(DEV)
    kw = dict(rowbias=rb.to(DEV), rows_per_batch=Fr * H * W, residual=rrows, split_k=1, tile=1 if ph == 8 else 2)
    assert o.conv_gn_fusable(rows, wp, n, H, W, kw["rowbias"], kw["rows_per_batch"])
    coef = o.group_norm_coeffs(rows, gd, bd, B, G, 1e-5)
    assert tuple(coef.shape) == (B, 2 * Cin)
    got, _, _ = o.conv3x3(rows, wp, bias.to(DEV), n, H, W, gn=(coef, Fr, silu), **kw)
    close(got, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype, scale=2.0)
    two, _, _ = o.conv3x3(o.group_norm(rows, gd, bd, B, G, 1e-5, silu), wp, bias.to(DEV), n, H, W, **kw)
    if dtype == torch.float32:
        close(got, two.float().cpu(), dtype)
    else:   # the same factors and arithmetic on the same inputs: a handful of elements may sit on a rounding boundary of the normalised
        # tensor (v_fma_f32 here, whatever the compiler contracts there): an output ulp on < 1e-3 of the elements
        d = (got.float() - two.float()).abs()
        frac, worst = float((d > 0).float().mean()), float(d.max()) / float(two.float().abs().max())
        assert frac < 1e-3 and worst <= 2 * torch.finfo(dtype).eps, (frac, worst)
    # a conv the halo kernel does not serve refuses the fusion instead of ignoring it
    from emote_hack_amd._lib import EmoHipError
    assert not o.conv_gn_fusable(rows, wp, n * (W // 8), H, 8)          # 8-pixel rows: the im2col loader
    with pytest.raises(EmoHipError):
        o.conv3x3(rows, wp, bias.to(DEV), n * (W // 8), H, 8, gn=(coef, Fr * (W // 8), silu), split_k=1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ph", [8, 16])
@pytest.mark.parametrize("n,H,W,Cin,Cout", [(3, 8, 8, 128, 192), (2, 16, 8, 64, 320), (5, 4, 16, 192, 132), (2, 12, 12, 64, 128)])
def test_conv3x3_halo_upsample(dtype, ph, n, H, W, Cin, Cout):
    """Upsample3D (resnet.py:74-82: F.interpolate(scale_factor=2, mode="nearest") -> conv 3x3) on the halo-reuse kernel: the
    patch grid lives on the upsampled frame and a halo pixel (y, x) is read from source pixel (y >> 1, x >> 1) - the upsampled
    tensor never exists.  Shapes whose upsampled frame is a multiple of the 8x16 / 16x16 patch; both patch heights."""
    if ph == 16 and (2 * H) % 16:
        pytest.skip("upsampled frame is not a multiple of the 16-row patch")
    o = ops()
    x = q(seeded_randn((n, Cin, H, W), 141), dtype)
    wt, bias = q(seeded_randn((Cout, Cin, 3, 3), 142) / math.sqrt(9 * Cin), dtype), 0.1 * seeded_randn((Cout,), 143)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, bias, padding=1)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
    wp = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    got, Ho, Wo = o.conv3x3(rows.to(DEV).to(dtype), wp.to(DEV).to(dtype).contiguous(), bias.to(DEV), n, H, W, upsample2x=True, split_k=1,
                            tile=1 if ph == 8 else 2)
    assert (Ho, Wo) == (2 * H, 2 * W)
    close(got, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("M,N,K,res", [(1000, 320, 320, True), (700, 640, 192, False), (513, 960, 64, True)])
def test_gemm_every_tile_shape(dtype, tile, M, N, K, res):
    """emo_gemm_params.tile pins 64x64 / 128x128 / 128x160 / 256x256 / 256x160 / 256x320 / 256x256 with the ping-pong main loop:
    every tile shape must produce the same GEMM (ragged M and N edges, residual epilogue)."""
    o = ops()
    a, w = q(seeded_randn((M, K), 112), dtype), q(seeded_randn((N, K), 113) / math.sqrt(K), dtype)
    bias, r = 0.1 * seeded_randn((N,), 114), q(seeded_randn((M, N), 115), dtype)
    ref = F.linear(a, w, bias) + (r if res else 0)
    got = o.gemm(a.to(DEV).to(dtype), w.to(DEV).to(dtype), bias.to(DEV), residual=r.to(DEV).to(dtype) if res else None, tile=tile, split_k=1)
    close(got, ref, dtype)


def _ln_fold(wt, bias, gamma, beta, dtype):
    """what unet._pack does: (W * gamma rounded to the compute dtype, its row sums, bias + W . beta)"""
    wp = (wt * gamma[None, :]).to(dtype)
    bp = wt.to(dtype).float() @ beta + (bias if bias is not None else 0)
    return wp, wp.float().sum(1), bp


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("M,N,K,mode", [(1000, 640, 320, "plain"), (520, 2560, 320, "geglu"), (768, 320, 320, "trans"),
                                        (2 * 3 * 64, 960, 320, "pe"), (300, 1280, 1280, "plain")])
def test_gemm_layernorm_fold(dtype, tile, M, N, K, mode):
    """LayerNorm folded into the GEMM (attention.py:279-316 LN -> Linear; motion_module.py:282-283 LN -> +PE -> Linear):
    out = LN(x) W^T + b, with a mean offset 2x the spread to exercise the colsum subtraction - plain / GEGLU / V^T epilogues, the
    temporal PE as a per-frame row bias, every tile shape."""
    o = ops()
    if mode == "geglu" and tile in (1, 3, 5, 6):
        pytest.skip("GEGLU pairs 32-column tiles (even tiles per wave): the planner falls back to 128x128")
    x = q(seeded_randn((M, K), 120) * 1.5 + 3.0 * seeded_randn((M, 1), 121), dtype)
    wt = seeded_randn((N, K), 122) / math.sqrt(K)
    bias = 0.1 * seeded_randn((N,), 123) if mode != "trans" else None
    gamma, beta = 1 + 0.2 * seeded_randn((K,), 124), 0.2 * seeded_randn((K,), 125)
    n = F.layer_norm(x, (K,), gamma, beta)
    wq = q(wt, dtype)
    kw, y = {}, None
    if mode == "pe":
        frames, rpf = 3, M // 6
        pe = seeded_randn((24, K), 126)
        fr = (torch.arange(M) // rpf) % frames
        y = F.linear(n + pe[fr], wq, bias)
        pe_w = (pe @ wq.t())[:frames].repeat(2, 1).contiguous()
        kw = dict(rowbias=pe_w.to(DEV), rows_per_batch=rpf)
    else:
        y = F.linear(n, wq, bias)
    if mode == "geglu":
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    wp, cs, bp = _ln_fold(wt, bias, gamma, beta, dtype)
    if mode == "geglu":   # (32 value, 32 gate) row interleave of unet._pack
        il = lambda t: torch.cat([t[:N // 2].reshape(N // 64, 32, *t.shape[1:]), t[N // 2:].reshape(N // 64, 32, *t.shape[1:])], 1).reshape(t.shape)
        wp, cs, bp = il(wp), il(cs), il(bp)
    a = x.to(DEV).to(dtype)
    st = o.layer_norm_stats(a, 1e-5)
    mu, var = x.mean(1), x.var(1, unbiased=False)
    torch.testing.assert_close(st.cpu(), torch.stack([mu, (var + 1e-5).rsqrt()], 1), rtol=1e-4, atol=1e-5)
    if mode == "trans":
        L = M // 3
        got = o.gemm(a, wp.to(DEV).contiguous(), bp.to(DEV), ln=(cs.to(DEV).contiguous(), st), transpose_rows=L, transpose_ld=L, tile=tile)
        got = got.float().cpu().permute(0, 2, 1).reshape(M, N)
    else:
        got = o.gemm(a, wp.to(DEV).contiguous(), bp.to(DEV).contiguous(), ln=(cs.to(DEV).contiguous(), st), geglu=(mode == "geglu"),
                     tile=tile, **kw)
    # the fold rounds W * gamma (not LN(x)) to the compute dtype: same error scale as the unfused pair
    tol = TOL[dtype]
    torch.testing.assert_close(got.float().cpu(), y, rtol=tol["rtol"], atol=tol["atol"] * (2.0 if dtype != torch.float32 else 1.0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("tile", [0, 2, 3, 4, 7])
@pytest.mark.parametrize("M,rpb,N,K", [(3 * 512, 512, 960, 320), (5 * 256 + 100, 256, 320, 192), (4 * 384, 384, 640, 64)])
def test_gemm_row_bias_uniform_over_a_tile(dtype, ln, tile, M, rpb, N, K):
    """The per-batch row bias (temb, resnet.py:188; the positional-encoding term of the temporal q|k|v projection,
    motion_module.py:246-248 folded to pe . W^T per frame) starts the accumulators when a tile's rows lie inside one batch
    (rows_per_batch a multiple of the tile height) and goes through the epilogue otherwise (rpb = 384 against 256-row tiles,
    a ragged last batch): both against bias + row bias in f32, with and without the LayerNorm fold (row bias divided by rstd)."""
    o = ops()
    x = q(seeded_randn((M, K), 190) * 1.3 + (1.5 * seeded_randn((M, 1), 191) if ln else 0), dtype)
    wt, bias = seeded_randn((N, K), 192) / math.sqrt(K), 0.1 * seeded_randn((N,), 193)
    nb = (M + rpb - 1) // rpb
    rb = seeded_randn((nb, N), 194)
    rows = torch.arange(M) // rpb
    a = x.to(DEV).to(dtype)
    if ln:
        gamma, beta = 1 + 0.2 * seeded_randn((K,), 195), 0.2 * seeded_randn((K,), 196)
        ref = F.linear(F.layer_norm(x, (K,), gamma, beta), q(wt, dtype), bias) + rb[rows]
        wp, cs, bp = _ln_fold(wt, bias, gamma, beta, dtype)
        got = o.gemm(a, wp.to(DEV).contiguous(), bp.to(DEV), ln=(cs.to(DEV).contiguous(), o.layer_norm_stats(a, 1e-5)), rowbias=rb.to(DEV),
                     rows_per_batch=rpb, tile=tile)
    else:
        ref = F.linear(x, q(wt, dtype), bias) + rb[rows]
        got = o.gemm(a, q(wt, dtype).to(DEV).to(dtype), bias.to(DEV), rowbias=rb.to(DEV), rows_per_batch=rpb, tile=tile, split_k=1)
    tol = TOL[dtype]
    torch.testing.assert_close(got.float().cpu(), ref, rtol=tol["rtol"], atol=tol["atol"] * (2.0 if ln and dtype != torch.float32 else 1.0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("M,L,N,K", [(512, 128, 200, 96), (1088, 64, 320, 320), (4096 + 64, 4096 + 64, 64, 64)])
def test_gemm_transposed_store_whole_tile_batches(dtype, ln, M, L, N, K):
    """V^T outputs whose batches are whole multiples of the 64-row tile (the UNet's case: L = H*W): ragged N, several batches, a
    batch longer than one tile, with and without the LayerNorm fold.  (An LDS-staged transposed store - full 128-byte lines per
    V^T row instead of 8-byte column pieces - was measured on these shapes: 79.0 -> 74.5 us at M=98304 N=K=320, nothing at the
    smaller levels; not kept.)"""
    o = ops()
    x = q(seeded_randn((M, K), 180) + (2.0 * seeded_randn((M, 1), 181) if ln else 0), dtype)
    wt, bias = seeded_randn((N, K), 182) / math.sqrt(K), 0.1 * seeded_randn((N,), 183)
    a = x.to(DEV).to(dtype)
    if ln:
        gamma, beta = 1 + 0.2 * seeded_randn((K,), 184), 0.2 * seeded_randn((K,), 185)
        ref = F.linear(F.layer_norm(x, (K,), gamma, beta), q(wt, dtype), bias)
        wp, cs, bp = _ln_fold(wt, bias, gamma, beta, dtype)
        got = o.gemm(a, wp.to(DEV).contiguous(), bp.to(DEV), ln=(cs.to(DEV).contiguous(), o.layer_norm_stats(a, 1e-5)), transpose_rows=L, transpose_ld=L)
    else:
        ref = F.linear(x, q(wt, dtype), bias)
        got = o.gemm(a, q(wt, dtype).to(DEV).to(dtype), bias.to(DEV), transpose_rows=L, transpose_ld=L)
    assert tuple(got.shape) == (M // L, N, L)
    tol = TOL[dtype]
    torch.testing.assert_close(got.float().cpu().permute(0, 2, 1).reshape(M, N), ref, rtol=tol["rtol"], atol=tol["atol"] * (2.0 if ln else 1.0))


def attn_ref(qh, k, v, scale):
    s = torch.matmul(qh, k.transpose(-1, -2)) * scale
    return torch.matmul(s.softmax(-1), v)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Lq,Lk,heads,d", [(2, 64, 64, 8, 40), (3, 100, 77, 4, 80), (2, 256, 256, 2, 160), (4, 16, 16, 4, 8),
                                             (1, 200, 333, 8, 16), (2, 130, 5, 8, 64)])
def test_attention_single_segment(dtype, B, Lq, Lk, heads, d):
    o = ops()
    C_ = heads * d
    qq, kk, vv = (q(seeded_randn((B, L, C_), s), dtype) for L, s in ((Lq, 24), (Lk, 25), (Lk, 26)))
    sp = lambda t: t.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    ref = attn_ref(sp(qq), sp(kk), sp(vv), d ** -0.5).permute(0, 2, 1, 3).reshape(B * Lq, C_)
    ld = (Lk + 7) // 8 * 8
    vt = torch.full((B, C_, ld), float("nan"))
    vt[:, :, :Lk] = vv.permute(0, 2, 1)
    got = o.attention(qq.reshape(-1, C_).to(DEV).to(dtype), kk.reshape(-1, C_).to(DEV).to(dtype), vt.to(DEV).to(dtype), Lk, B=B,
                      Lq=Lq, heads=heads, d=d, scale=d ** -0.5)
    close(got, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Lq,Lk,heads,d,div", [(24, 4096 - 100, 77, 8, 40, 12), (48, 1024, 77, 8, 80, 1), (130, 256 + 17, 50, 8, 160, 1),
                                                 (40, 1000, 128, 8, 40, 4)])
def test_attention_resident_context(dtype, B, Lq, Lk, heads, d, div):
    """Cross-attention to a short context (attention.py:300-305; 77 text keys): every KV tile fits the ring, so a block keeps
    them and walks SEVERAL 128-query tiles of its (b, head) (attention.hip, resident mode - picked by the launch when there are
    enough blocks left).  Ragged last q tile (a block's walk ends early), ragged key tile, context shared by `div` batch rows."""
    o = ops()
    C_ = heads * d
    qq = q(seeded_randn((B, Lq, C_), 41), dtype)
    kk, vv = (q(seeded_randn((B // div, Lk, C_), s_), dtype) for s_ in (42, 43))
    sp = lambda t: t.reshape(t.shape[0], -1, heads, d).permute(0, 2, 1, 3)
    ref = attn_ref(sp(qq), sp(kk).repeat_interleave(div, 0), sp(vv).repeat_interleave(div, 0), d ** -0.5).permute(0, 2, 1, 3).reshape(B * Lq, C_)
    ld = (Lk + 7) // 8 * 8
    vt = torch.full((B // div, C_, ld), float("nan"))
    vt[:, :, :Lk] = vv.permute(0, 2, 1)
    got = o.attention(qq.reshape(-1, C_).to(DEV).to(dtype), kk.reshape(-1, C_).to(DEV).to(dtype), vt.to(DEV).to(dtype), Lk, B=B,
                      Lq=Lq, heads=heads, d=d, scale=d ** -0.5, seg0_div=div)
    close(got, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_bank_segment_and_shared_context(dtype):
    """Reference read path: K/V = cat([x, bank repeated over F]) for c rows, plain self-attention for
    uc rows (mutual_self_attention.py:238-256); text context shared by the F frames (attention.py:118)."""
    o = ops()
    Bc, Fr, L, heads, d, Lb = 2, 3, 48, 4, 40, 80
    C_ = heads * d
    nb = Bc * Fr
    qq, kk, vv = (q(seeded_randn((nb, L, C_), s), dtype) for s in (27, 28, 29))
    bk, bv = q(seeded_randn((Bc, Lb, C_), 30), dtype), q(seeded_randn((Bc, Lb, C_), 31), dtype)
    sp = lambda t: t.reshape(t.shape[0], -1, heads, d).permute(0, 2, 1, 3)
    refs = []
    for b in range(nb):
        k_, v_ = kk[b:b + 1], vv[b:b + 1]
        if b >= Fr:  # second half = conditional rows
            k_ = torch.cat([k_, bk[b // Fr:b // Fr + 1]], 1)
            v_ = torch.cat([v_, bv[b // Fr:b // Fr + 1]], 1)
        refs.append(attn_ref(sp(qq[b:b + 1]), sp(k_), sp(v_), d ** -0.5).permute(0, 2, 1, 3).reshape(L, C_))
    ref = torch.cat(refs)
    dv = lambda t: t.to(DEV).to(dtype)
    got = o.attention(dv(qq.reshape(-1, C_)), dv(kk.reshape(-1, C_)), dv(vv.permute(0, 2, 1).contiguous()), L, B=nb, Lq=L,
                      heads=heads, d=d, scale=d ** -0.5, k1=dv(bk.reshape(-1, C_)), v1t=dv(bv.permute(0, 2, 1).contiguous()),
                      Lk1=Lb, seg1_div=Fr, seg1_first_batch=Fr)
    close(got, ref, dtype)
    # cond-only bank (seg1_skip): the uncond bank row is never read, so a bank that holds the cond rows only gives the same
    got_c = o.attention(dv(qq.reshape(-1, C_)), dv(kk.reshape(-1, C_)), dv(vv.permute(0, 2, 1).contiguous()), L, B=nb, Lq=L,
                        heads=heads, d=d, scale=d ** -0.5, k1=dv(bk[1:].reshape(-1, C_)), v1t=dv(bv[1:].permute(0, 2, 1).contiguous()),
                        Lk1=Lb, seg1_div=Fr, seg1_first_batch=Fr, seg1_skip=1)
    assert torch.equal(got_c, got)
    # shared context: batch b reads context row b // Fr
    ck, cv = q(seeded_randn((Bc, 7, C_), 32), dtype), q(seeded_randn((Bc, 7, C_), 33), dtype)
    ref2 = torch.cat([attn_ref(sp(qq[b:b + 1]), sp(ck[b // Fr:b // Fr + 1]), sp(cv[b // Fr:b // Fr + 1]), d ** -0.5)
                      .permute(0, 2, 1, 3).reshape(L, C_) for b in range(nb)])
    vt = torch.zeros(Bc, C_, 8)
    vt[:, :, :7] = cv.permute(0, 2, 1)
    got2 = o.attention(dv(qq.reshape(-1, C_)), dv(ck.reshape(-1, C_)), dv(vt), 7, B=nb, Lq=L, heads=heads, d=d, scale=d ** -0.5,
                       seg0_div=Fr)
    close(got2, ref2, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [40, 48])
def test_attention_two_query_tiles_per_wave(dtype, d):
    """The d <= 48 kernel of the 2-byte types with TWO 32-query tiles per wave (256 query rows per block; picked when the launch
    has >= 1024 such blocks: 8 batches x 8 heads x 4096 queries here - the 64x64 level's shape).  K rows are stored UNPADDED at
    d = 40 (5 chunks: the last k-step's second half reads the next row against a zero Q chunk) and padded at d = 48 (7).
    Ragged key counts in both segments (200 = 3 tiles + 8 keys, 77 = 1 tile + 13), a bank row selected by the device word for
    batches >= 4, a late spike that forces the O rescale, and a ragged last query block are all on this path."""
    o = ops()
    B, Lq, Lk, Lb, heads = 8, 4096 - 40, 200, 77, 8
    C_ = heads * d
    qq, kk, vv = (q(seeded_randn((B, L, C_), s), dtype) for L, s in ((Lq, 44), (Lk, 45), (Lk, 46)))
    bk, bv = q(seeded_randn((3, Lb, C_), 47), dtype), q(seeded_randn((3, Lb, C_), 48), dtype)
    kk[5, 150] = q(qq[5, 300] * 3.0, dtype)
    row = 2
    sp = lambda t: t.reshape(t.shape[0], -1, heads, d).permute(0, 2, 1, 3)
    dvf = lambda t: t.to(DEV).float()
    refs = []
    for b in range(B):
        k_, v_ = kk[b:b + 1], vv[b:b + 1]
        if b >= 4:
            k_, v_ = torch.cat([k_, bk[row:row + 1]], 1), torch.cat([v_, bv[row:row + 1]], 1)
        refs.append(attn_ref(sp(dvf(qq[b:b + 1])), sp(dvf(k_)), sp(dvf(v_)), d ** -0.5).permute(0, 2, 1, 3).reshape(Lq, C_).cpu())
    ref = torch.cat(refs)
    dv = lambda t: t.to(DEV).to(dtype)
    vt = torch.full((B, C_, 200), float("nan"))
    vt[:, :, :Lk] = vv.permute(0, 2, 1)
    bvt = torch.zeros(3, C_, 80)
    bvt[:, :, :Lb] = bv.permute(0, 2, 1)
    got = o.attention(dv(qq.reshape(-1, C_)), dv(kk.reshape(-1, C_)), dv(vt), Lk, B=B, Lq=Lq, heads=heads, d=d, scale=d ** -0.5,
                      k1=dv(bk.reshape(-1, C_)), v1t=dv(bvt), Lk1=Lb, seg1_div=B, seg1_first_batch=4,
                      seg1_row=torch.tensor([row], dtype=torch.int32, device=DEV))
    close(got, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d,per_tile_log2", [(40, 7.0), (40, 9.0), (80, 7.0)])
def test_attention_thresholded_rescale_score_ramp(dtype, d, per_tile_log2):
    """The worst case of the THRESHOLDED online softmax (attention.hip ATT_TAU = 8 log2 units): every row's maximum climbs by
    `per_tile_log2` per 64-key tile over many tiles.  At 7 the running reference never moves after the first tile of a run, so P
    reaches 2^7..2^8 before the 2-byte pack and the denominator row accumulates those values; at 9 every tile takes the rescale
    branch.  Both against an f32 softmax of the same (quantised) inputs: the bound of the un-rescaled path is pinned on its own,
    not only through the model-level tests."""
    o = ops()
    B, Lq, Lk, heads = 2, 256, 64 * 24 + 13, 2
    C_ = heads * d
    scale = d ** -0.5
    qq = 0.25 * seeded_randn((B, Lq, C_), 61)
    kk = 0.25 * seeded_randn((B, Lk, C_), 62)
    vv = seeded_randn((B, Lk, C_), 63)
    # one channel per head carries the ramp: q = A, k_j = ramp_j / (A * scale) -> score_j = ramp_j (+ the small random part)
    A = 4.0
    ramp = (torch.arange(Lk, dtype=torch.float32) / 64.0) * per_tile_log2 * math.log(2.0)
    for h in range(heads):
        qq[:, :, h * d] = A
        kk[:, :, h * d] = ramp / (A * scale)
    qq, kk, vv = q(qq, dtype), q(kk, dtype), q(vv, dtype)
    sp = lambda t: t.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    ref = attn_ref(sp(qq).double(), sp(kk).double(), sp(vv).double(), scale).float().permute(0, 2, 1, 3).reshape(B * Lq, C_)
    ld = (Lk + 7) // 8 * 8
    vt = torch.zeros(B, C_, ld)
    vt[:, :, :Lk] = vv.permute(0, 2, 1)
    dv = lambda t: t.to(DEV).to(dtype)
    got = o.attention(dv(qq.reshape(-1, C_)), dv(kk.reshape(-1, C_)), dv(vt), Lk, B=B, Lq=Lq, heads=heads, d=d, scale=scale)
    close(got, ref, dtype)
    # ... and with the ramp continuing through a second (bank) segment
    got2 = o.attention(dv(qq.reshape(-1, C_)), dv(kk[:, :640].reshape(-1, C_)), dv(vt[:, :, :640].contiguous()), 640, B=B, Lq=Lq, heads=heads, d=d,
                       scale=scale, k1=dv(kk[:, 640:].reshape(-1, C_)), v1t=dv(vt[:, :, 640:].contiguous()), Lk1=Lk - 640, seg1_div=1,
                       seg1_first_batch=0)
    close(got2, ref, dtype)


def test_attention_online_softmax_rescale_forced():
    """Spike one key late in the sequence so the running max jumps in a later KV tile (rescale branch)."""
    o = ops()
    B, L, heads, d = 1, 192, 1, 64
    qq, kk, vv = seeded_randn((B, L, d), 34), seeded_randn((B, L, d), 35), seeded_randn((B, L, d), 36)
    kk[0, 150] = qq[0, 10] * 4.0
    ref = attn_ref(qq[:, None], kk[:, None], vv[:, None], d ** -0.5).reshape(L, d)
    got = o.attention(qq.reshape(-1, d).to(DEV), kk.reshape(-1, d).to(DEV), vv.permute(0, 2, 1).contiguous().to(DEV), L, B=B, Lq=L,
                      heads=heads, d=d, scale=d ** -0.5)
    close(got, ref, torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Fr,HW,heads,d", [(2, 4, 16, 4, 16), (1, 12, 64, 8, 40), (2, 12, 4, 8, 160), (1, 24, 9, 8, 80)])
def test_temporal_attention(dtype, B, Fr, HW, heads, d):
    o = ops()
    C_ = heads * d
    qkv = q(seeded_randn((B * Fr * HW, 3 * C_), 37), dtype)
    t = qkv.reshape(B, Fr, HW, 3, heads, d).permute(3, 0, 2, 4, 1, 5)  # (3, B, HW, heads, F, d)
    ref = attn_ref(t[0], t[1], t[2], d ** -0.5)  # (B, HW, heads, F, d)
    ref = ref.permute(0, 3, 1, 2, 4).reshape(B * Fr * HW, C_)
    got = o.temporal_attention(qkv.to(DEV).to(dtype), B, Fr, HW, heads, d, d ** -0.5)
    close(got, ref, dtype)


def test_cfg_step_and_accumulate():
    o = ops()
    from oracle.scheduler_ref import SchedulerRef, counter_normal
    C4, Ft, HW = 4, 6, 20
    npred = seeded_randn((2, C4, Ft, HW), 38)
    counter = torch.tensor([1.0, 2, 1, 3, 1, 2])
    lat = seeded_randn((C4 * Ft * HW,), 39)
    for kind, t in (("ddim", 501), ("ddpm", 500), ("ddpm", 0)):
        sch = SchedulerRef(kind)
        sch.set_timesteps(50)
        cx, ce, cn = sch.coefficients(t)
        avg = npred / counter.view(1, 1, Ft, 1)
        eps = avg[0] + 7.5 * (avg[1] - avg[0])
        z = counter_normal(3, 7, lat.numel())
        ref = cx * lat + ce * eps.reshape(-1) + cn * z
        l_dev, e_dev = lat.clone().to(DEV), torch.empty(lat.numel(), device=DEV)
        o.cfg_step(npred.to(DEV), counter.to(DEV), l_dev, C_=C4, F=Ft, HW=HW, guidance_scale=7.5, c_x=cx, c_eps=ce, c_noise=cn,
                   seed=3, step=7, eps_out=e_dev)
        torch.testing.assert_close(e_dev.cpu(), eps.reshape(-1), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(l_dev.cpu(), ref, rtol=1e-4, atol=1e-4)
    # window accumulate
    pred = seeded_randn((3 * HW, C4), 40)
    acc = torch.zeros(C4, Ft, HW, device=DEV)
    cnt = torch.zeros(Ft, device=DEV)
    frames = torch.tensor([4, 5, 0], dtype=torch.int32, device=DEV)
    o.accumulate_window(pred.to(DEV), acc, cnt, frames, C_=C4, F=Ft, HW=HW, add_counter=True)
    ref = torch.zeros(C4, Ft, HW)
    ref[:, [4, 5, 0]] = pred.reshape(3, HW, C4).permute(2, 0, 1)
    torch.testing.assert_close(acc.cpu(), ref)
    assert cnt.cpu().tolist() == [1, 0, 0, 0, 1, 1]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sk", [2, 5])
def test_gemm_split_k_matches_single_pass(dtype, sk):
    """split-K (small-M levels) must reproduce the fused single-pass epilogue: bias, rowbias, GEGLU,
    residual, scale and the V^T store, with a deterministic fixed-order reduction."""
    o = ops()
    M, K, No = 2 * 48, 640, 64
    a = q(seeded_randn((M, K), 41), dtype)
    w, bias = q(seeded_randn((2 * No, K), 42) / 25, dtype), 0.1 * seeded_randn((2 * No,), 43)
    res = q(seeded_randn((M, 2 * No), 44), dtype)
    rb = seeded_randn((2, 2 * No), 45)
    dv = lambda t: t.to(DEV).to(dtype)
    ref = (F.linear(a, w, bias) + rb.repeat_interleave(48, 0) + res) * 0.5
    got = o.gemm(dv(a), dv(w), bias.to(DEV), rowbias=rb.to(DEV), rows_per_batch=48, residual=dv(res), out_scale=0.5, split_k=sk)
    close(got, ref, dtype)
    got_b = o.gemm(dv(a), dv(w), bias.to(DEV), rowbias=rb.to(DEV), rows_per_batch=48, residual=dv(res), out_scale=0.5, split_k=sk)
    assert torch.equal(got, got_b)  # deterministic
    hv, gate = F.linear(a, w, bias).chunk(2, dim=-1)
    wi = torch.cat([w[:No].reshape(No // 32, 32, K), w[No:].reshape(No // 32, 32, K)], 1).reshape(2 * No, K)
    bi = torch.cat([bias[:No].reshape(No // 32, 32), bias[No:].reshape(No // 32, 32)], 1).reshape(2 * No)
    close(o.gemm(dv(a), dv(wi).contiguous(), bi.to(DEV).contiguous(), geglu=True, split_k=sk), hv * F.gelu(gate), dtype)
    got3 = o.gemm(dv(a), dv(w), transpose_rows=48, transpose_ld=48, split_k=sk)
    close(got3, F.linear(a, w).reshape(2, 48, 2 * No).permute(0, 2, 1), dtype)
    # conv with split-K (8x8 level of the ReferenceNet: M=128)
    x = q(seeded_randn((2, 64, 8, 8), 46), dtype)
    wt = q(seeded_randn((96, 64, 3, 3), 47) / 24, dtype)
    refc = F.conv2d(x, wt, None, padding=1).permute(0, 2, 3, 1).reshape(-1, 96)
    gotc, _, _ = o.conv3x3(dv(x.permute(0, 2, 3, 1).reshape(-1, 64).contiguous()), dv(wt.permute(0, 2, 3, 1).reshape(96, 576)).contiguous(),
                           None, 2, 8, 8, split_k=sk)
    close(gotc, refc, dtype)
