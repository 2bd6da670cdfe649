"""Cross-attention to the 77-key text context: q tiles per block (EMO_ATT_QREP = 1 is the one-tile-per-block launch; the override
exists only in a sweep build: TUS=attention tools/bench/build_variant.sh att_qrep -DEMO_ATT_QREP_ENV, EMO_HIP_LIB=.../att_qrep.so)."""
import os
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from emote_hack_amd import ops as o  # noqa: E402
from gemm_tiles import timeit, NROT  # noqa: E402
dev, dt = "cuda", torch.bfloat16
print("EMO_ATT_QREP =", os.environ.get("EMO_ATT_QREP", "(planned)"))
for B, Lq, heads, d in ((24, 4096, 8, 40), (24, 1024, 8, 80), (24, 256, 8, 160), (24, 64, 8, 160)):
    C = heads * d
    qs = [torch.randn(B * Lq, C, device=dev, dtype=dt) for _ in range(NROT)]
    k = torch.randn(2 * 77, C, device=dev, dtype=dt)
    vt = torch.randn(2, C, 80, device=dev, dtype=dt)
    us = timeit(lambda i: o.attention(qs[i % NROT], k, vt, 77, B=B, Lq=Lq, heads=heads, d=d, scale=d ** -0.5, seg0_div=B // 2))
    print(f"B={B} Lq={Lq} d={d}: {us:7.1f} us  ({2 * B * Lq * C * 2 / us / 1e3:.0f} GB/s q+out)", flush=True)
