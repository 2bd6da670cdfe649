"""One attention shape, a few launches (for rocprofv3 --pmc): python tools/bench/attn_pmc.py [B Lq Lk0 Lk1 heads d first]"""
import sys
import torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
a = [int(x) for x in sys.argv[1:]] or [24, 4096, 4096, 4096, 8, 40, 12]
B, L, Lk0, Lk1, heads, d, first = a
dev, dt = 'cuda', torch.bfloat16
C = heads * d
q = torch.randn(B * L, C, device=dev, dtype=dt); k = torch.randn(B * Lk0, C, device=dev, dtype=dt)
vt = torch.randn(B, C, (Lk0 + 7) // 8 * 8, device=dev, dtype=dt)
kw = {}
if Lk1:
    kw = dict(k1=torch.randn(2 * Lk1, C, device=dev, dtype=dt), v1t=torch.randn(2, C, Lk1, device=dev, dtype=dt), Lk1=Lk1, seg1_div=B,
              seg1_first_batch=first, seg1_row=torch.zeros(1, dtype=torch.int32, device=dev))
for _ in range(3):
    o.attention(q, k, vt, Lk0, B=B, Lq=L, heads=heads, d=d, scale=d ** -0.5, **kw)
torch.cuda.synchronize()
