"""debug: GroupNorm+SiLU inside the halo conv vs the unfused pair, with a centre-tap identity weight (out = normalised input)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'


def case(dtype, B, Fr, H, W, Cin, tap=4, ph=None, silu=True):
    n, G = B * Fr, 32
    torch.manual_seed(0)
    x = (torch.randn(n * H * W, Cin) * 1.5 + 0.3).to(dev).to(dtype)
    g, b = (1 + 0.1 * torch.randn(Cin)).to(dev), (0.1 * torch.randn(Cin)).to(dev)
    Cout = Cin
    w = torch.zeros(Cout, 9, Cin)
    w[torch.arange(Cout), tap, torch.arange(Cin)] = 1.0
    wp = w.reshape(Cout, 9 * Cin).to(dev).to(dtype).contiguous()
    kw = dict(split_k=1)
    if ph: kw["tile"] = 1 if ph == 8 else 2
    coef = o.group_norm_coeffs(x, g, b, B, G, 1e-5)
    got, _, _ = o.conv3x3(x, wp, None, n, H, W, gn=(coef, Fr, silu), **kw)
    xn = o.group_norm(x, g, b, B, G, 1e-5, silu)
    two, _, _ = o.conv3x3(xn, wp, None, n, H, W, **kw)
    torch.cuda.synchronize()
    d = (got.float() - two.float()).abs()
    bad = d > 1e-3
    print(f"{dtype} B={B} Fr={Fr} {H}x{W} Cin={Cin} tap={tap} ph={ph}: max diff {float(d.max()):.4g}, bad {int(bad.sum())} / {bad.numel()}, finite {bool(got.isfinite().all())}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("   bad rows (img,y,x):", [(int(r) // (H * W), int(r) % (H * W) // W, int(r) % W) for r in rows[:24]], "n_rows", len(rows))
        print("   bad cols:", cols[:64].tolist(), "n_cols", len(cols))
        r0 = int(rows[0])
        print("   row", r0, "got", got[r0, :8].float().tolist(), "want", two[r0, :8].float().tolist(), "raw", x[r0, :8].float().tolist())
        # coefficient check
        c = coef[0].view(-1, 4)
        print("   coef pair0", c[0].tolist())


for dt in (torch.float32, torch.bfloat16):
    bk = 32 if dt == torch.float32 else 64
    case(dt, 1, 1, 8, 16, bk, ph=8)
    case(dt, 1, 1, 8, 16, bk, tap=0, ph=8)
    case(dt, 1, 1, 8, 16, 2 * bk, ph=8)
    case(dt, 1, 1, 16, 16, bk, ph=16)
    case(dt, 1, 1, 16, 32, 2 * bk, ph=8)
    case(dt, 2, 3, 16, 32, 4 * bk, ph=8)
    case(dt, 2, 3, 16, 32, 4 * bk, ph=16)
