"""Attention microbenchmark over the path's shapes (bf16): python tools/bench/attn_bench.py [lib.so ...]
Each library (default: the product build; variants from tools/bench/build_variant.sh) runs in its own process
(EMO_HIP_LIB selects it).  FLOPs are counted per batch row: rows below seg1_first_batch read the first segment only."""
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
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16


def run(B, L, Lk0, Lk1, heads, d, first=0):
    C = heads * d
    q = torch.randn(B * L, C, device=dev, dtype=dt); k = torch.randn(B * Lk0, C, device=dev, dtype=dt)
    vt = torch.randn(B, C, (Lk0 + 7) // 8 * 8, device=dev, dtype=dt)
    kw = {}
    if Lk1:
        kw = dict(k1=torch.randn(2 * Lk1, C, device=dev, dtype=dt), v1t=torch.randn(2, C, Lk1, device=dev, dtype=dt), Lk1=Lk1, seg1_div=B,
                  seg1_first_batch=first, seg1_row=torch.zeros(1, dtype=torch.int32, device=dev))
    f = lambda: o.attention(q, k, vt, Lk0, B=B, Lq=L, heads=heads, d=d, scale=d ** -0.5, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 4.0 * heads * L * d * (B * Lk0 + (B - first) * Lk1)
    print(f"B={B} Lq={L} Lk={Lk0}+{Lk1} (bank from batch {first}) h={heads} d={d}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s", flush=True)


# the 64x64 level (d = 40): mid/up blocks under CFG (12 uncond + 12 cond frames), down blocks, cross attention, ReferenceNet pass
run(24, 4096, 4096, 4096, 8, 40, first=12); run(24, 4096, 4096, 4096, 8, 40); run(24, 4096, 4096, 0, 8, 40); run(24, 4096, 77, 0, 8, 40)
run(10, 4096, 4096, 0, 8, 40); run(2, 4096, 4096, 0, 8, 40)
run(24, 1024, 1024, 1024, 8, 80, first=12); run(24, 1024, 1024, 0, 8, 80); run(24, 1024, 77, 0, 8, 80)
run(24, 256, 256, 256, 8, 160, first=12); run(24, 256, 77, 0, 8, 160); run(24, 64, 64, 64, 8, 160, first=12)
