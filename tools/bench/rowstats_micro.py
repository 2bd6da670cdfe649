"""Producer / consumer pairs of the LayerNorm-statistics hand-over (emo_gemm_params.row_stats / ln_slots) in isolation:
producer with and without the statistics epilogue, consumer fed by the (mean, rstd) table of emo_layernorm_stats vs the slot table,
and the chain producer -> [stats kernel] -> consumer as one graph.  python tools/bench/rowstats_micro.py"""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16


def timed(f, n=10):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(M, C, N2, geglu):
    att = torch.randn(M, C, device=dev, dtype=dt); wo = torch.randn(C, C, device=dev, dtype=dt) / 30; bo = torch.randn(C, device=dev)
    res = torch.randn(M, C, device=dev, dtype=dt)
    w2 = torch.randn(N2, C, device=dev, dtype=dt) / 30; b2 = torch.randn(N2, device=dev); cs = w2.float().sum(1).contiguous()
    h, st = o.gemm(att, wo, bo, residual=res, row_stats_eps=1e-5)
    stk = o.layer_norm_stats(h, 1e-5)
    t_p0 = timed(lambda: o.gemm(att, wo, bo, residual=res))
    t_p1 = timed(lambda: o.gemm(att, wo, bo, residual=res, row_stats_eps=1e-5))
    t_k = timed(lambda: o.layer_norm_stats(h, 1e-5))
    t_c0 = timed(lambda: o.gemm(h, w2, b2, ln=(cs, stk), geglu=geglu))
    t_c1 = timed(lambda: o.gemm(h, w2, b2, ln=(cs, st), geglu=geglu))

    def chain0():
        hh = o.gemm(att, wo, bo, residual=res)
        return o.gemm(hh, w2, b2, ln=(cs, o.layer_norm_stats(hh, 1e-5)), geglu=geglu)

    def chain1():
        hh, s_ = o.gemm(att, wo, bo, residual=res, row_stats_eps=1e-5)
        return o.gemm(hh, w2, b2, ln=(cs, s_), geglu=geglu)
    t_ch0, t_ch1 = timed(chain0), timed(chain1)
    print(f"M={M:6d} C={C:5d} N2={N2:5d} slots={st.slots:3d}: producer {t_p0:6.1f} -> {t_p1:6.1f} us | stats kernel {t_k:5.1f} | consumer {t_c0:6.1f} -> {t_c1:6.1f} | "
          f"chain {t_ch0:6.1f} -> {t_ch1:6.1f} us", flush=True)


run(98304, 320, 2560, True)
run(98304, 320, 960, False)
run(24576, 640, 5120, True)
run(24576, 640, 1920, False)
run(6144, 1280, 10240, True)
run(6144, 1280, 3840, False)
run(1536, 1280, 10240, True)
