"""Run-to-run determinism + accuracy of emo_attention over small / ragged shapes (debug aid)."""
import sys, itertools
import torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'
bad = 0
for dt in (torch.bfloat16, torch.float32):
    for (B, L, Lk, heads, d) in [(3, 256, 256, 4, 8), (3, 64, 64, 4, 16), (3, 16, 16, 4, 16), (3, 4, 4, 4, 16), (3, 256, 5, 4, 8), (3, 64, 5, 4, 16),
                                 (2, 256, 320, 4, 8), (2, 1024, 1024, 8, 40), (2, 300, 700, 8, 40), (24, 4096, 4096, 8, 40), (4, 1024, 1024, 8, 80)]:
        C = heads * d
        torch.manual_seed(0)
        q = torch.randn(B * L, C, device=dev).to(dt); k = torch.randn(B * Lk, C, device=dev).to(dt)
        ld = (Lk + 7) // 8 * 8
        vt = torch.full((B, C, ld), float('nan'), device=dev).to(dt)
        v = torch.randn(B, Lk, C, device=dev).to(dt)
        vt[:, :, :Lk] = v.permute(0, 2, 1)
        outs = []
        for rep in range(4):
            # dirty the LDS / caches between runs with another shape
            if rep % 2: o.attention(torch.randn(2 * 512, 320, device=dev).to(dt) * 50, torch.randn(2 * 512, 320, device=dev).to(dt) * 50, torch.randn(2, 320, 512, device=dev).to(dt) * 1e4, 512, B=2, Lq=512, heads=8, d=40, scale=1.0)
            outs.append(o.attention(q, k, vt, Lk, B=B, Lq=L, heads=heads, d=d, scale=d ** -0.5).clone())
        same = all(torch.equal(outs[0], x) for x in outs[1:])
        sp = lambda t, n: t.reshape(B, n, heads, d).permute(0, 2, 1, 3).float()
        ref = torch.softmax(sp(q, L) @ sp(k, Lk).transpose(-1, -2) * d ** -0.5, -1) @ sp(v.reshape(B * Lk, C), Lk)
        ref = ref.permute(0, 2, 1, 3).reshape(B * L, C)
        err = float((outs[0].float() - ref).abs().max())
        fin = bool(torch.isfinite(outs[0]).all())
        flag = "" if (same and fin and err < (2e-2 if dt == torch.bfloat16 else 1e-4)) else "   <<<<<< BAD"
        bad += bool(flag)
        print(f"{str(dt):16s} B={B} Lq={L} Lk={Lk} h={heads} d={d}: deterministic={same} finite={fin} max|err|={err:.2e}{flag}", flush=True)
print("BAD" if bad else "ALL OK", bad)
