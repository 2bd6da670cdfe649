"""The path's heaviest GEMM / conv launches, a few times each (subject of tools/bench/path_pmc.sh under rocprofv3 --pmc).
Every launch has its own grid size, so the counter rows can be told apart by (kernel, grid)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16
def ln_args(M, K, N, geglu=False):
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    cs, b = w.float().sum(1).contiguous(), torch.randn(N, device=dev)
    return x, w, b, (cs, o.layer_norm_stats(x, 1e-5))
def dense(M, N, K, geglu=False, ln=False, res=False):
    if ln:
        x, w, b, lnp = ln_args(M, K, N)
        f = lambda: o.gemm(x, w, b, geglu=geglu, ln=lnp)
    else:
        x = torch.randn(M, K, device=dev, dtype=dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt); b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev, dtype=dt) if res else None
        f = lambda: o.gemm(x, w, b, residual=r)
    for _ in range(3): f()
def conv(n_img, H, W, Cin, Cout, res=False):
    x = torch.randn(n_img * H * W, Cin, device=dev, dtype=dt); w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
    b = torch.randn(Cout, device=dev); r = torch.randn(n_img * H * W, Cout, device=dev, dtype=dt) if res else None
    for _ in range(3): o.conv3x3(x, w, b, n_img, H, W, residual=r)
dense(98304, 2560, 320, geglu=True, ln=True)
dense(24576, 5120, 640, geglu=True, ln=True)
dense(98304, 320, 1600, res=True)
dense(98304, 320, 320, res=True)
dense(98304, 960, 320, ln=True)
dense(24576, 640, 640, res=True)
conv(24, 64, 64, 320, 320)
conv(24, 32, 32, 640, 640)
conv(24, 16, 16, 1280, 1280)
torch.cuda.synchronize()
