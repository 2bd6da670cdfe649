"""3x3 conv with / without the in-conv GroupNorm + SiLU over the resnet shapes (bf16): python tools/bench/gn_conv_bench.py [lib.so ...]
(each library in its own process; EMO_HIP_LIB selects it).  'gn' = conv3x3(gn=...) on raw rows; 'plain' = the same conv on a
pre-normalised tensor; 'apply' = the emo_groupnorm_apply pass the fusion removes (statistics excluded: they remain either way)."""
import os, subprocess, sys

if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "product":
            env["EMO_HIP_LIB"] = os.path.abspath(lib)
        print(f"=== {lib}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
    sys.exit(0)

import torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o, _lib
dev = 'cuda'; dt = torch.bfloat16


def timed(f, reps=12):
    for i in range(3): f(i)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(0)
        with torch.cuda.graph(g, stream=s):
            for i in range(reps): f(i)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(B, Fr, H, W, Cin, N, nrot=4):
    n = B * Fr
    xs = [torch.randn(n * H * W, Cin, device=dev, dtype=dt) for _ in range(nrot)]
    w = torch.randn(N, 9 * Cin, device=dev, dtype=dt) / 50; b = torch.randn(N, device=dev)
    g_, b_ = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    rs = [torch.randn(n * H * W, N, device=dev, dtype=dt) for _ in range(nrot)]
    coef = o.group_norm_coeffs(xs[0], g_, b_, B, 32, 1e-5)
    lib = _lib.load()
    S = Fr * H * W
    part = torch.empty(max(lib.emo_groupnorm_workspace_bytes(B, S, Cin, 32) // 4, 1), device=dev, dtype=torch.float32)
    st = lambda: torch.cuda.current_stream().cuda_stream
    lib.emo_groupnorm_stats(xs[0].data_ptr(), Cin, part.data_ptr(), B, S, Cin, 32, 1, st())
    ys = [torch.empty_like(x) for x in xs]
    t_gn = timed(lambda i: o.conv3x3(xs[i % nrot], w, b, n, H, W, residual=rs[i % nrot], gn=(coef, Fr, True)))
    t_pl = timed(lambda i: o.conv3x3(xs[i % nrot], w, b, n, H, W, residual=rs[i % nrot]))
    t_ap = timed(lambda i: lib.emo_groupnorm_apply(xs[i % nrot].data_ptr(), Cin, part.data_ptr(), g_.data_ptr(), b_.data_ptr(), ys[i % nrot].data_ptr(), Cin,
                                                  B, S, Cin, 32, 1e-5, 1, 1, st()))
    M = n * H * W
    fl = 2.0 * M * N * 9 * Cin
    print(f"M={M:6d} N={N:5d} Cin={Cin:5d}: gn {t_gn:7.1f} us {fl / t_gn / 1e6:6.0f} TF/s | plain {t_pl:7.1f} us {fl / t_pl / 1e6:6.0f} TF/s | apply {t_ap:6.1f} us | "
          f"fused - (plain + apply) = {t_gn - t_pl - t_ap:+7.1f} us", flush=True)


run(2, 12, 64, 64, 320, 320)
run(2, 12, 64, 64, 640, 320)
run(2, 12, 64, 64, 960, 320)
run(2, 12, 32, 32, 640, 640)
run(2, 12, 32, 32, 1280, 640)
run(2, 12, 16, 16, 1280, 1280)
run(2, 12, 16, 16, 2560, 1280)
