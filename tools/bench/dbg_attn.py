"""f32 / low-precision attention vs a torch statement at the full-size shapes of BASELINE configs[4] (768x768: 9216 queries)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'


def run(dt, B, Lq, Lk, heads, d):
    C = heads * d
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(B * Lq, C, device=dev, generator=g).to(dt); k = torch.randn(B * Lk, C, device=dev, generator=g).to(dt)
    v = torch.randn(B, Lk, C, device=dev, generator=g).to(dt)
    ld = (Lk + 7) // 8 * 8
    vt = torch.full((B, C, ld), float('nan'), device=dev, dtype=dt)
    vt[:, :, :Lk] = v.permute(0, 2, 1)
    got = o.attention(q, k, vt, Lk, B=B, Lq=Lq, heads=heads, d=d, scale=d ** -0.5).float()
    sp = lambda t, L: t.float().reshape(B, L, heads, d).permute(0, 2, 1, 3)
    ref = torch.softmax(sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) * d ** -0.5, -1) @ sp(v, Lk)
    ref = ref.permute(0, 2, 1, 3).reshape(B * Lq, C)
    print(f"{dt} B={B} Lq={Lq} Lk={Lk} d={d}: nan={int(torch.isnan(got).sum())} max err {float((got - ref).abs().nan_to_num(9e9).max()):.3e}", flush=True)


for dt in (torch.float32, torch.float16, torch.bfloat16):
    run(dt, 4, 9216, 9216, 8, 40); run(dt, 48, 9216, 5, 8, 40); run(dt, 4, 2304, 2304, 8, 80); run(dt, 48, 2304, 5, 8, 80)
    run(dt, 4, 576, 576, 8, 160); run(dt, 48, 576, 5, 8, 160); run(dt, 48, 144, 144, 8, 160); run(dt, 48, 144, 5, 8, 160)
