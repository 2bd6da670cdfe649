"""Temporal attention over the path's shapes (bf16): us per launch and algorithmic GB/s (qkv read + out written)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16
for (B, F, HW, heads, d) in [(2, 12, 4096, 8, 40), (2, 12, 1024, 8, 80), (2, 12, 256, 8, 160), (2, 12, 64, 8, 160), (1, 12, 4096, 8, 40), (2, 24, 4096, 8, 40)]:
    C = heads * d
    qkv = torch.randn(B * F * HW, 3 * C, device=dev, dtype=dt)
    f = lambda: o.temporal_attention(qkv, B, F, HW, heads, d, d ** -0.5)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B} F={F} HW={HW} h={heads} d={d}: {us:7.1f} us  {B * F * HW * 4 * C * 2 / us / 1e3:7.0f} GB/s", flush=True)
