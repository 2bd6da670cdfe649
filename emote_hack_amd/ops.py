"""Thin host wrappers: torch tensors (device memory, streams = plumbing) -> raw pointers -> C ABI.

Every function launches hand-written HIP kernels from libemo_hip.so on torch's current stream.
Nothing here computes with torch; tensors are only allocated (`torch.empty`) and addressed.
Activations are 2-D "rows x channels" tensors (NHWC rows), possibly views with a leading
dimension (`.stride(0)`) larger than their width.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import EMO_BF16, EMO_F16, EMO_F32, AttentionParams, GemmParams, check

_DT = {torch.float32: EMO_F32, torch.bfloat16: EMO_BF16, torch.float16: EMO_F16}


def dt(t_or_dtype) -> int:
    d = t_or_dtype if isinstance(t_or_dtype, torch.dtype) else t_or_dtype.dtype
    try:
        return _DT[d]
    except KeyError:
        raise _lib.EmoHipError(f"unsupported compute dtype {d} (float32 | bfloat16 | float16)")


def vec(dtype) -> int:
    return 4 if dtype == torch.float32 else 8


class KernelProfiler:
    """Live per-kernel timing with HIP events on the launch stream (bench.py `roofline`).  Each profiled
    launch is bracketed by two events recorded on the stream the kernel is launched on; durations and
    the launch's ALGORITHMIC flops / bytes are summed per kernel class after a synchronise."""

    def __init__(self):
        self.records = []   # (name, flops, bytes, ev0, ev1)

    def launch(self, name, flops, nbytes, fn, tag=""):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.append((name, flops, nbytes, e0, e1, tag))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, fl, nb, e0, e1, _tag in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out

    def by_shape(self):
        torch.cuda.synchronize()
        out = {}
        for name, fl, nb, e0, e1, tag in self.records:
            d = out.setdefault((name, tag), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


PROFILER = None  # set to a KernelProfiler to time launches


def _launch(name, flops, nbytes, fn, tag=""):
    if PROFILER is None:
        fn()
    else:
        PROFILER.launch(name, flops, nbytes, fn, tag)


def _stream():
    if not torch.cuda.is_available():
        raise _lib.EmoHipError("no HIP device: emote_hack_amd ops launch gfx950 kernels (there is no CPU fallback)")
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EmoHipError("emote_hack_amd ops need device tensors (no CPU fallback on the product path)")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _rows(t):
    """(ptr, ld) of a 2-D row-major view."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return _ptr(t), t.stride(0)


# ----------------------------------------------------------------------------- layout / elementwise
def ncfhw_to_rows(x5d: torch.Tensor, dtype, cpad=None) -> torch.Tensor:
    _need_cuda(x5d)
    B, Cc, F, H, W = x5d.shape
    x5d = x5d.contiguous().float()
    cpad = cpad or Cc
    y = torch.empty(B * F * H * W, cpad, device=x5d.device, dtype=dtype)
    check(_lib.load().emo_ncfhw_to_rows(_ptr(x5d), _ptr(y), B, Cc, F, H, W, cpad, cpad, dt(dtype), _stream()), "emo_ncfhw_to_rows")
    return y


def rows_to_ncfhw(x: torch.Tensor, B, Cc, F, H, W) -> torch.Tensor:
    _need_cuda(x)
    p, ld = _rows(x)
    y = torch.empty(B, Cc, F, H, W, device=x.device, dtype=torch.float32)
    check(_lib.load().emo_rows_to_ncfhw(p, _ptr(y), B, Cc, F, H, W, ld, dt(x), _stream()), "emo_rows_to_ncfhw")
    return y


def copy_cols(x: torch.Tensor, y: torch.Tensor, coff: int):
    px, ldx = _rows(x)
    py, ldy = _rows(y)
    check(_lib.load().emo_copy_cols(px, ldx, py, ldy, coff, x.shape[0], x.shape[1], dt(x), _stream()), "emo_copy_cols")


def concat_cols(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=channels) (unet_3d_blocks.py:629,731) as two strided row copies."""
    y = torch.empty(a.shape[0], a.shape[1] + b.shape[1], device=a.device, dtype=a.dtype)
    copy_cols(a, y, 0)
    copy_cols(b, y, a.shape[1])
    return y


def add(a: torch.Tensor, b: torch.Tensor, alpha=1.0, out=None) -> torch.Tensor:
    out = torch.empty_like(a) if out is None else out
    pa, lda = _rows(a)
    pb, ldb = _rows(b)
    po, ldo = _rows(out)
    check(_lib.load().emo_add(pa, lda, pb, ldb, float(alpha), po, ldo, a.shape[0], a.shape[1], dt(a), _stream()), "emo_add")
    return out


def convert(src: torch.Tensor, dtype, fp16_round=False) -> torch.Tensor:
    _need_cuda(src)
    src = src.contiguous()
    dst = torch.empty(src.shape, device=src.device, dtype=dtype)
    check(_lib.load().emo_convert(_ptr(src), dt(src), _ptr(dst), dt(dtype), src.numel(), int(fp16_round), _stream()), "emo_convert")
    return dst


def silu(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.load().emo_silu(_ptr(x), _ptr(y), x.numel(), dt(x), _stream()), "emo_silu")
    return y


def timestep_embedding(timesteps: torch.Tensor, freqs: torch.Tensor, dim: int, flip: bool, dtype) -> torch.Tensor:
    _need_cuda(timesteps, freqs)
    assert timesteps.dtype == torch.int64 and freqs.dtype == torch.float32
    B = timesteps.shape[0]
    out = torch.empty(B, dim, device=timesteps.device, dtype=dtype)
    check(_lib.load().emo_timestep_embedding(_ptr(timesteps), _ptr(freqs), _ptr(out), B, dim, int(flip), dt(dtype), _stream()),
          "emo_timestep_embedding")
    return out


# ----------------------------------------------------------------------------- norms
GN_ONE_LAUNCH = True   # A/B hook (bench.py --gn-two-launch): False keeps every GroupNorm on the stats + apply pair


def group_norm(x: torch.Tensor, gamma, beta, n_inst: int, groups: int, eps: float, silu_: bool, out=None) -> torch.Tensor:
    """GroupNorm over NHWC rows; x (n_inst*S, C).  n_inst=B -> joint 5-D statistics (resnet.py:180),
    n_inst=B*F -> per frame (attention.py:124)."""
    _need_cuda(x)
    lib = _lib.load()
    M, Cc = x.shape
    S = M // n_inst
    px, ldx = _rows(x)
    y = torch.empty(M, Cc, device=x.device, dtype=x.dtype) if out is None else out
    py, ldy = _rows(y)
    if GN_ONE_LAUNCH and lib.emo_groupnorm_one_launch_ok(n_inst, S, Cc, groups, dt(x)):   # small instances: one launch, one read
        _launch("groupnorm", 0.0, x.element_size() * 2.0 * M * Cc,
                lambda: check(lib.emo_groupnorm(px, ldx, _ptr(gamma), _ptr(beta), py, ldy, n_inst, S, Cc, groups, float(eps), int(silu_),
                                                dt(x), _stream()), "emo_groupnorm"), tag=f"M={M} C={Cc}{' silu' if silu_ else ''} 1L")
        return y
    ws = lib.emo_groupnorm_workspace_bytes(n_inst, S, Cc, groups)
    part = torch.empty(max(ws // 4, 1), device=x.device, dtype=torch.float32)

    def run():
        check(lib.emo_groupnorm_stats(px, ldx, _ptr(part), n_inst, S, Cc, groups, dt(x), _stream()), "emo_groupnorm_stats")
        check(lib.emo_groupnorm_apply(px, ldx, _ptr(part), _ptr(gamma), _ptr(beta), py, ldy, n_inst, S, Cc, groups, float(eps),
                                      int(silu_), dt(x), _stream()), "emo_groupnorm_apply")
    _launch("groupnorm", 0.0, x.element_size() * 2.0 * M * Cc, run, tag=f"M={M} C={Cc}{' silu' if silu_ else ''}")   # algorithmic: one read + one write
    return y


def group_norm_coeffs(x: torch.Tensor, gamma, beta, n_inst: int, groups: int, eps: float) -> torch.Tensor:
    """The statistics half of a GroupNorm whose normalisation runs inside its consumer (conv3x3(gn=...)): one read-only pass over x,
    then the per-(instance, channel) factors (n_inst, 2C) f32, channel pairs interleaved (scale, scale, shift, shift) - emo_hip.h
    emo_groupnorm_coeffs.  The normalised tensor is never materialised."""
    _need_cuda(x)
    lib = _lib.load()
    M, Cc = x.shape
    S = M // n_inst
    px, ldx = _rows(x)
    part = torch.empty(max(lib.emo_groupnorm_workspace_bytes(n_inst, S, Cc, groups) // 4, 1), device=x.device, dtype=torch.float32)
    coef = torch.empty(n_inst, 2 * Cc, device=x.device, dtype=torch.float32)

    def run():
        check(lib.emo_groupnorm_stats(px, ldx, _ptr(part), n_inst, S, Cc, groups, dt(x), _stream()), "emo_groupnorm_stats")
        check(lib.emo_groupnorm_coeffs(_ptr(part), _ptr(gamma), _ptr(beta), _ptr(coef), n_inst, S, Cc, groups, float(eps), dt(x), _stream()),
              "emo_groupnorm_coeffs")
    _launch("groupnorm_stats", 0.0, x.element_size() * 1.0 * M * Cc, run, tag=f"M={M} C={Cc}")   # algorithmic: one read
    return coef


def group_norm_fold_linear(x: torch.Tensor, gamma, beta, n_inst: int, groups: int, eps: float, w: torch.Tensor, bias):
    """GroupNorm (no activation) folded into the Linear / 1x1 conv behind it (emo_hip.h emo_groupnorm_fold_linear): one
    statistics pass over x, then per-instance weights (n_inst, Cout, C) and a per-instance bias (n_inst, Cout) f32 for
    gemm(x, w_n, bias_n, w_slab_rows=S) - the normalised tensor is never materialised."""
    _need_cuda(x, w)
    lib = _lib.load()
    M, Cc = x.shape
    S = M // n_inst
    Cout = w.shape[0]
    assert w.shape == (Cout, Cc) and w.is_contiguous() and w.dtype == x.dtype
    px, ldx = _rows(x)
    part = torch.empty(max(lib.emo_groupnorm_workspace_bytes(n_inst, S, Cc, groups) // 4, 1), device=x.device, dtype=torch.float32)
    wn = torch.empty(n_inst, Cout, Cc, device=x.device, dtype=x.dtype)
    rb = torch.empty(n_inst, Cout, device=x.device, dtype=torch.float32)

    def run():
        check(lib.emo_groupnorm_stats(px, ldx, _ptr(part), n_inst, S, Cc, groups, dt(x), _stream()), "emo_groupnorm_stats")
        check(lib.emo_groupnorm_fold_linear(_ptr(part), _ptr(gamma), _ptr(beta), _ptr(w), _ptr(bias), _ptr(wn), _ptr(rb), n_inst, S, Cc, groups,
                                            Cout, float(eps), dt(x), _stream()), "emo_groupnorm_fold_linear")
    _launch("groupnorm_fold", 0.0, x.element_size() * 1.0 * M * Cc, run, tag=f"M={M} C={Cc}")   # algorithmic: one read
    return wn, rb


def layer_norm(x: torch.Tensor, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0) -> torch.Tensor:
    _need_cuda(x)
    M, Cc = x.shape
    px, ldx = _rows(x)
    y = torch.empty(M, Cc, device=x.device, dtype=x.dtype)
    _launch("layernorm", 0.0, x.element_size() * 2.0 * M * Cc,
            lambda: check(_lib.load().emo_layernorm(px, ldx, _ptr(gamma), _ptr(beta), _ptr(y), Cc, M, Cc, float(eps), _ptr(pe),
                                                    rows_per_frame, frames, dt(x), _stream()), "emo_layernorm"))
    return y


def layer_norm_stats(x: torch.Tensor, eps=1e-5) -> torch.Tensor:
    """(mean, rstd) per row, f32 (M, 2): the statistics pass of a LayerNorm folded into the consumer GEMM (gemm(ln=...))."""
    _need_cuda(x)
    M, Cc = x.shape
    px, ldx = _rows(x)
    st = torch.empty(M, 2, device=x.device, dtype=torch.float32)
    _launch("layernorm_stats", 0.0, x.element_size() * 1.0 * M * Cc,
            lambda: check(_lib.load().emo_layernorm_stats(px, ldx, _ptr(st), M, Cc, float(eps), dt(x), _stream()), "emo_layernorm_stats"))
    return st


# ----------------------------------------------------------------------------- GEMM / conv
_VT_VERDICT = {}   # (M, w shape, vt geometry, dtype, tile hint) -> emo_gemm_vt_ok's answer: unsupported levels go straight to two launches
GEMM_TILE = 0   # tuning hook for tools/bench: pins emo_gemm_params.tile of every dense GEMM that does not pass tile=


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, *, rowbias=None, rows_per_batch=0, residual=None, geglu=False,
         out_scale=1.0, out=None, transpose_rows=0, transpose_ld=0, conv=None, split_k=None, ln=None, tile=None,
         w_slab_rows=0, gn=None, vt_cols=0, vt_rows=0, vt_ld=0):
    """out = epilogue(a @ w.T).  a (M, K) rows view; w (N, K) contiguous in the compute dtype.
    ln = (colsum f32 (N,), stats f32 (M, 2) from layer_norm_stats(a)): LayerNorm over K folded into the GEMM - a holds the RAW
    rows, w / bias carry the folded affine (emo_hip.h emo_gemm_params.ln_colsum).
    conv = dict(H, W, Cin, stride, upsample2x, Ho, Wo) selects the implicit 3x3 conv loader (then a is
    the (B*F*H*W, >=Cin) NHWC input and M = B*F*Ho*Wo).
    transpose_rows=L stores V^T per batch of L rows: out (M/L, N, transpose_ld).
    gn = (coef from group_norm_coeffs(a), images per instance, silu): GroupNorm (+ SiLU) of the conv's input applied inside the
    halo-reuse 3x3 conv - a holds the RAW rows (emo_hip.h emo_gemm_params.gn_coef; conv_gn_fusable says which convs qualify).
    vt_cols = n: the LAST n output columns are stored transposed per batch of vt_rows rows (V^T (M / vt_rows, n, vt_ld)), the first N - n
    row-major: returns (out (M, N - n), vt) - or None when this GEMM is not served that way (emo_hip.h emo_gemm_params.vt; the caller
    then runs two launches)."""
    _need_cuda(a, w)
    if vt_cols and _VT_VERDICT.get((a.shape[0], tuple(w.shape), vt_cols, vt_rows, vt_ld, a.dtype, tile, GEMM_TILE)) is False:
        return None     # asked before (emo_gemm_vt_ok below): not served - no throw-away allocations, no ctypes call
    p = GemmParams()
    pa, lda = _rows(a)
    if w_slab_rows:     # per-instance weights (n_inst, N, K) and bias (n_inst, N): rows [i * w_slab_rows, ...) use slab i (group_norm_fold_linear)
        assert w.dim() == 3 and conv is None and ln is None and a.shape[0] == w.shape[0] * w_slab_rows, (w.shape, a.shape, w_slab_rows)
        assert bias is None or (bias.shape == w.shape[:2] and bias.is_contiguous()), bias.shape
        p.w_slab_rows, p.w_slab_stride = w_slab_rows, w.stride(0)
        split_k = 1
    N, K = w.shape[-2:]
    assert w.is_contiguous() and w.dtype == a.dtype, (w.dtype, a.dtype)
    if conv is None:
        M = a.shape[0]
        assert a.shape[1] == K, (a.shape, w.shape)
    else:
        M = (a.shape[0] // (conv["H"] * conv["W"])) * conv["Ho"] * conv["Wo"]
    n_out = N // 2 if geglu else N
    if transpose_rows:
        nb = M // transpose_rows
        if out is None:
            out = torch.empty(nb, n_out, transpose_ld, device=a.device, dtype=a.dtype)   # pad columns [L, ld) are cleaned by the attention loader
        p.transpose_out, p.t_rows, p.t_ld, p.t_batch_stride = 1, transpose_rows, transpose_ld, n_out * transpose_ld
        p.C, p.ldc = out.data_ptr(), 0
    else:
        if out is None:
            out = torch.empty(M, n_out - vt_cols, device=a.device, dtype=a.dtype)
        pc, ldc = _rows(out)
        p.C, p.ldc = pc.value, ldc
    p.A, p.lda, p.W = pa.value, lda, w.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.stride(1) == 1
        p.rowbias, p.rows_per_batch, p.ld_rowbias = rowbias.data_ptr(), rows_per_batch, rowbias.stride(0)
    if residual is not None:
        pr, ldr = _rows(residual)
        p.residual, p.ldr = pr.value, ldr
    p.M, p.N, p.K = M, N, K
    p.geglu, p.out_scale = int(geglu), float(out_scale)
    if conv is not None:
        p.conv_taps, p.H, p.W_, p.Cin = 9, conv["H"], conv["W"], conv["Cin"]
        p.stride, p.upsample2x, p.Ho, p.Wo = conv["stride"], int(conv["upsample2x"]), conv["Ho"], conv["Wo"]
        p.conv_asym = int(conv.get("asym", 0))
        p.up_h, p.up_w = conv.get("up", (0, 0))
    p.dtype = dt(a)
    vt = None
    if vt_cols:
        assert conv is None and not geglu and residual is None and not transpose_rows and tuple(out.shape) == (M, N - vt_cols)
        vt = torch.empty(M // vt_rows, vt_cols, vt_ld, device=a.device, dtype=a.dtype)
        p.vt, p.vt_col0, p.t_rows, p.t_ld, p.t_batch_stride = vt.data_ptr(), N - vt_cols, vt_rows, vt_ld, vt_cols * vt_ld
    if gn is not None:
        coef, imgs_per_inst, gn_silu = gn
        assert conv is not None and coef.dtype == torch.float32 and coef.is_contiguous() and coef.shape[1] == 2 * conv["Cin"], coef.shape
        p.gn_coef, p.gn_imgs_per_inst, p.gn_silu = coef.data_ptr(), int(imgs_per_inst), int(bool(gn_silu))
        split_k = 1
    p.tile = int(tile if tile is not None else (GEMM_TILE if conv is None else 0))
    if ln is not None:
        colsum, stats = ln
        assert conv is None and colsum.dtype == torch.float32 and colsum.numel() == N
        assert stats.dtype == torch.float32 and stats.shape == (M, 2) and stats.is_contiguous()
        p.ln_colsum, p.ln_stats = colsum.data_ptr(), stats.data_ptr()
        split_k = 1   # the correction rides in the accumulator init of ONE pass over K
    lib = _lib.load()
    if vt_cols:
        split_k = 1
        ok = bool(lib.emo_gemm_vt_ok(C.byref(p)))
        if out.data_ptr() % 16 == 0 and vt.data_ptr() % 16 == 0 and out.stride(0) == N - vt_cols:   # (the verdict of a plain geometry only)
            _VT_VERDICT[(a.shape[0], tuple(w.shape), vt_cols, vt_rows, vt_ld, a.dtype, tile, GEMM_TILE)] = ok
        if not ok:
            return None
    sk = lib.emo_gemm_suggest_split_k(M, N, K, p.dtype, int(bool(geglu)), int(bool(transpose_rows))) if split_k is None else split_k
    ws = None
    if sk > 1:
        ws = torch.empty(lib.emo_gemm_workspace_bytes(M, N, sk) // 4, device=a.device, dtype=torch.float32)
        p.split_k, p.workspace = sk, ws.data_ptr()
    esz = a.element_size()
    _launch("gemm_conv3x3" if conv is not None else "gemm_dense", 2.0 * M * N * K,
            esz * (float(M) * (K if conv is None else conv["Cin"]) + float(N) * K + float(M) * n_out),
            lambda: check(_lib.load().emo_gemm(C.byref(p), _stream()), "emo_gemm"),
            tag=f"M={M} N={N} K={K}" + (" geglu" if geglu else "") + (" T" if transpose_rows else "") +
                (f" s{conv['stride']}{'u' if conv['upsample2x'] else ''}" if conv is not None else "") + (f" sk{sk}" if sk > 1 else "") +
                (" ln" if ln is not None else "") + (" slab" if w_slab_rows else "") + (" rowbias" if rowbias is not None else "") +
                (" gn" if gn is not None else "") + (f" vt{vt_cols}" if vt_cols else ""))
    return (out, vt) if vt_cols else out


def conv_gn_fusable(x: torch.Tensor, w: torch.Tensor, n_img: int, H: int, W: int, rowbias=None, rows_per_batch=0) -> bool:
    """Whether the stride-1 3x3 conv of x (n_img*H*W, >= Cin) with the re-laid weight w runs on the halo-reuse kernel, i.e. may take
    its GroupNorm (+ SiLU) along as conv3x3(gn=...) (emo_hip.h emo_conv3x3_gn_fusable)."""
    p = GemmParams()
    _, lda = _rows(x)
    p.lda, p.M, p.N, p.K = lda, n_img * H * W, w.shape[0], w.shape[1]
    p.conv_taps, p.H, p.W_, p.Cin, p.stride, p.Ho, p.Wo = 9, H, W, w.shape[1] // 9, 1, H, W
    p.dtype = dt(x)
    if rowbias is not None:
        p.rowbias, p.rows_per_batch, p.ld_rowbias = rowbias.data_ptr(), rows_per_batch, rowbias.stride(0)
    return bool(_lib.load().emo_conv3x3_gn_fusable(C.byref(p)))


def conv3x3(x: torch.Tensor, w: torch.Tensor, bias, n_img: int, H: int, W: int, *, stride=1, upsample2x=False, pad=1, upsample_to=None, **kw):
    """Per-frame 3x3 conv, pad 1 (resnet.py:30-38), as an implicit GEMM over NHWC rows.
    w is the re-laid (Cout, 9*Cin_pad) weight.  pad=0 means the asymmetric (0, 1, 0, 1) padding of the VAE encoder's downsampler."""
    cin = w.shape[1] // 9
    if upsample_to is not None and tuple(upsample_to) == (2 * H, 2 * W):
        upsample2x, upsample_to = True, None        # the x2 fast path (a shift instead of a division per tap)
    He, We = (2 * H, 2 * W) if upsample2x else (tuple(upsample_to) if upsample_to is not None else (H, W))
    tot = 2 if pad == 1 else 1
    Ho, Wo = (He + tot - 3) // stride + 1, (We + tot - 3) // stride + 1
    assert x.shape[0] == n_img * H * W and pad in (0, 1)
    return gemm(x, w, bias, conv=dict(H=H, W=W, Cin=cin, stride=stride, upsample2x=upsample2x, Ho=Ho, Wo=Wo, asym=int(pad == 0),
                                      up=(He, We) if upsample_to is not None else (0, 0)), **kw), Ho, Wo


# ----------------------------------------------------------------------------- attention
def attention(q, k0, v0t, Lk0, *, B, Lq, heads, d, scale, seg0_div=1, k1=None, v1t=None, Lk1=0, seg1_div=1,
              seg1_first_batch=0, seg1_skip=0, seg1_row=None) -> torch.Tensor:
    """q (B*Lq, >=heads*d) rows view; k0 rows view; v0t (Bk, heads*d, ld) V^T tensors.
    seg1_row: device int32 tensor holding the bank row every batch >= seg1_first_batch reads (see emo_hip.h)."""
    _need_cuda(q, k0, v0t)
    p = AttentionParams()
    pq, ldq = _rows(q)
    pk, ldk = _rows(k0)
    out = torch.empty(B * Lq, heads * d, device=q.device, dtype=q.dtype)
    p.q, p.ldq = pq.value, ldq
    p.k0, p.ldk0, p.v0t, p.ldv0t, p.Lk0 = pk.value, ldk, v0t.data_ptr(), v0t.stride(1), Lk0
    p.seg0_div = seg0_div
    if k1 is not None:
        pk1, ldk1 = _rows(k1)
        p.k1, p.ldk1, p.v1t, p.ldv1t, p.Lk1 = pk1.value, ldk1, v1t.data_ptr(), v1t.stride(1), Lk1
        p.seg1_div, p.seg1_first_batch, p.seg1_skip = seg1_div, seg1_first_batch, seg1_skip
        if seg1_row is not None:
            assert seg1_row.dtype == torch.int32 and seg1_row.is_cuda
            p.seg1_row = seg1_row.data_ptr()
    else:
        p.seg1_div = 1
    p.out, p.ldo = out.data_ptr(), out.stride(0)
    p.B, p.Lq, p.heads, p.d, p.scale, p.dtype = B, Lq, heads, d, float(scale), dt(q)
    esz = q.element_size()
    # (only the batch rows from seg1_first_batch on read the second segment - under CFG the uncond half does not)
    b1 = max(B - seg1_first_batch, 0) if k1 is not None else 0
    _launch("attention", 4.0 * heads * Lq * d * (B * Lk0 + b1 * Lk1), esz * heads * d * (2.0 * B * Lq + 2.0 * (B * Lk0 + b1 * Lk1)),
            lambda: check(_lib.load().emo_attention(C.byref(p), _stream()), "emo_attention"),
            tag=f"B={B} Lq={Lq} Lk={Lk0}+{Lk1} h={heads} d={d}")
    return out


def temporal_attention(qkv: torch.Tensor, B, F, HW, heads, d, scale) -> torch.Tensor:
    _need_cuda(qkv)
    pq, ld = _rows(qkv)
    out = torch.empty(B * F * HW, heads * d, device=qkv.device, dtype=qkv.dtype)
    _launch("temporal_attention", 4.0 * B * HW * heads * F * F * d, qkv.element_size() * 4.0 * B * F * HW * heads * d,
            lambda: check(_lib.load().emo_temporal_attention(pq, ld, _ptr(out), out.stride(0), B, F, HW, heads, d, float(scale),
                                                             dt(qkv), _stream()), "emo_temporal_attention"),
            tag=f"B={B} F={F} HW={HW} h={heads} d={d}")
    return out


# ----------------------------------------------------------------------------- sampler
def cfg_step(noise_pred, counter, latents, *, C_, F, HW, guidance_scale, c_x, c_eps, c_noise, seed, step, eps_out=None):
    _need_cuda(noise_pred, counter, latents)
    assert noise_pred.dtype == counter.dtype == latents.dtype == torch.float32
    check(_lib.load().emo_cfg_step(_ptr(noise_pred), _ptr(counter), _ptr(latents), _ptr(eps_out), C_, F, HW, float(guidance_scale),
                                   float(c_x), float(c_eps), float(c_noise), int(seed) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF,
                                   _stream()), "emo_cfg_step")


def accumulate_window(pred_rows, noise_pred_branch, counter, frames_i32, *, C_, F, HW, add_counter):
    pp, ld = _rows(pred_rows)
    check(_lib.load().emo_accumulate_window(pp, ld, _ptr(noise_pred_branch), _ptr(counter), _ptr(frames_i32), frames_i32.numel(),
                                            C_, F, HW, int(add_counter), dt(pred_rows), _stream()), "emo_accumulate_window")


# ----------------------------------------------------------------------------- EMO conditioning ops (A17/A18)
_ACT = {"silu": 0, "relu": 1, "tanh": 2, "gelu": 3}


def act(x: torch.Tensor, kind: str) -> torch.Tensor:
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.load().emo_act(_ptr(x), _ptr(y), x.numel(), _ACT[kind], dt(x), _stream()), "emo_act")
    return y


def speed_encode(v: torch.Tensor, centers: torch.Tensor, radii: torch.Tensor, dtype) -> torch.Tensor:
    _need_cuda(v, centers, radii)
    out = torch.empty(v.shape[0], centers.shape[0], device=v.device, dtype=dtype)
    check(_lib.load().emo_speed_encode(_ptr(v), _ptr(centers), _ptr(radii), _ptr(out), v.shape[0], centers.shape[0], dt(dtype),
                                       _stream()), "emo_speed_encode")
    return out


def speed_bucket(v: torch.Tensor, centers: torch.Tensor) -> torch.Tensor:
    _need_cuda(v, centers)
    idx = torch.empty(v.shape[0], device=v.device, dtype=torch.int32)
    check(_lib.load().emo_speed_bucket(_ptr(v), _ptr(centers), _ptr(idx), v.shape[0], centers.shape[0], _stream()), "emo_speed_bucket")
    return idx


def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    out = torch.empty(idx.shape[0], table.shape[1], device=table.device, dtype=table.dtype)
    check(_lib.load().emo_gather_rows(_ptr(table), _ptr(idx), _ptr(out), idx.shape[0], table.shape[1], table.shape[0], dt(table),
                                      _stream()), "emo_gather_rows")
    return out


def add_rowbias(x: torch.Tensor, rb: torch.Tensor, rows_per_batch: int) -> torch.Tensor:
    px, ldx = _rows(x)
    pr, ldr = _rows(rb)
    y = torch.empty(x.shape[0], x.shape[1], device=x.device, dtype=x.dtype)
    check(_lib.load().emo_add_rowbias(px, ldx, pr, ldr, _ptr(y), y.stride(0), x.shape[0], x.shape[1], rows_per_batch, dt(x), _stream()),
          "emo_add_rowbias")
    return y


# ----------------------------------------------------------------------------- either side of the loop (VAE, audio front-end)
def softmax_rows(x: torch.Tensor, scale: float, out=None) -> torch.Tensor:
    """softmax(scale * x) over the columns of a rows view (M, N)."""
    _need_cuda(x)
    px, ldx = _rows(x)
    y = torch.empty(x.shape[0], x.shape[1], device=x.device, dtype=x.dtype) if out is None else out
    py, ldy = _rows(y)
    check(_lib.load().emo_softmax_rows(px, ldx, py, ldy, x.shape[0], x.shape[1], float(scale), dt(x), _stream()), "emo_softmax_rows")
    return y


def audio_windows(feats: torch.Tensor, m: int = 2, n: int = 2) -> torch.Tensor:
    """(T, D) -> (T, m+n+1, D): features of frames [t-m, t+n], zero-padded at the ends (Net.py:649-667)."""
    _need_cuda(feats)
    feats = feats.contiguous()
    T_, D = feats.shape
    out = torch.empty(T_, m + n + 1, D, device=feats.device, dtype=feats.dtype)
    check(_lib.load().emo_audio_windows(_ptr(feats), _ptr(out), T_, D, m, n, dt(feats), _stream()), "emo_audio_windows")
    return out


def rows_to_video(x: torch.Tensor, B, Cc, F, H, W, mul=0.5, add=0.5, lo=0.0, hi=1.0) -> torch.Tensor:
    """rows ((b f) h w, >= C) -> (B, C, F, H, W) f32 = clamp(x*mul + add, lo, hi) (EMOAnimationPipeline.py:303-306)."""
    _need_cuda(x)
    px, ld = _rows(x)
    y = torch.empty(B, Cc, F, H, W, device=x.device, dtype=torch.float32)
    check(_lib.load().emo_rows_to_video(px, ld, _ptr(y), B, Cc, F, H * W, float(mul), float(add), float(lo), float(hi), dt(x), _stream()),
          "emo_rows_to_video")
    return y


def channel_norm(x: torch.Tensor, gamma, beta, eps=1e-5, gelu=False) -> torch.Tensor:
    """nn.GroupNorm(C, C) over a sequence: x rows (S, C) normalised per channel over the S rows, affine, optional erf-GELU (emo_channelnorm)."""
    _need_cuda(x)
    lib = _lib.load()
    S, Cc = x.shape
    px, ldx = _rows(x)
    y = torch.empty(S, Cc, device=x.device, dtype=x.dtype)
    ws = torch.empty(max(lib.emo_channelnorm_workspace_bytes(S, Cc) // 4, 1), device=x.device, dtype=torch.float32)
    _launch("channelnorm", 0.0, x.element_size() * 3.0 * S * Cc,
            lambda: check(lib.emo_channelnorm(px, ldx, _ptr(gamma), _ptr(beta), _ptr(y), Cc, S, Cc, float(eps), int(bool(gelu)), _ptr(ws), dt(x), _stream()),
                          "emo_channelnorm"))
    return y


def maxpool2x2(x: torch.Tensor, n_img: int, H: int, W: int) -> torch.Tensor:
    """nn.MaxPool2d(2, 2) over NHWC rows (Net.py:828)."""
    _need_cuda(x)
    px, ldx = _rows(x)
    y = torch.empty(n_img * (H // 2) * (W // 2), x.shape[1], device=x.device, dtype=x.dtype)
    check(_lib.load().emo_maxpool2x2(px, ldx, _ptr(y), y.stride(0), n_img, H, W, x.shape[1], dt(x), _stream()), "emo_maxpool2x2")
    return y


def bilinear_to_nchw(x: torch.Tensor, n_img: int, Cc: int, h: int, w: int, Ho: int, Wo: int) -> torch.Tensor:
    """F.interpolate(size=(Ho, Wo), mode='bilinear', align_corners=False) of NHWC rows into (n, C, Ho, Wo) f32 (Net.py:851)."""
    _need_cuda(x)
    px, ld = _rows(x)
    y = torch.empty(n_img, Cc, Ho, Wo, device=x.device, dtype=torch.float32)
    check(_lib.load().emo_bilinear_to_nchw(px, ld, _ptr(y), n_img, Cc, h, w, Ho, Wo, dt(x), _stream()), "emo_bilinear_to_nchw")
    return y
