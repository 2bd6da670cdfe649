"""Temporal attention over the path's shapes (bf16): us per launch and algorithmic GB/s (qkv read + out written)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16
for (B, F, HW, heads, d) in [(2, 12, 4096, 8, 40), (2, 12, 1024, 8, 80), (2, 12, 256, 8, 160), (2, 12, 64, 8, 160), (1, 12, 4096, 8, 40), (2, 24, 4096, 8, 40)]:
    C = heads * d
    # one buffer re-read every launch sits in the Infinity Cache (hot); rotating over enough of them to exceed its 256 MB is what the step sees (cold)
    nrot = max(2, int(600e6 // (B * F * HW * 3 * C * 2)) + 1)
    qs = [torch.randn(B * F * HW, 3 * C, device=dev, dtype=dt) for _ in range(nrot)]
    res = []
    for rot in (1, nrot):
        f = lambda i: o.temporal_attention(qs[i % rot], B, F, HW, heads, d, d ** -0.5)
        for i in range(3): f(i)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): f(i)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    gb = B * F * HW * 4 * C * 2 / 1e3
    print(f"B={B} F={F} HW={HW} h={heads} d={d}: hot {res[0]:7.1f} us {gb / res[0]:6.0f} GB/s | cold ({nrot} buffers) {res[1]:7.1f} us {gb / res[1]:6.0f} GB/s", flush=True)
