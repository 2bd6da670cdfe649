"""3x3 halo conv: which operand is behind the fabric-side traffic in excess of the algorithmic bytes?  Shapes that isolate the
activation stream (big A, tiny W), the weight stream (tiny A, big W, incl. weights beyond the 256 MB Infinity Cache) and the
path's own level-0 / level-2 convs; 3 launches each, in this order (tools/bench/conv_traffic.sh reads the counters per launch)."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16
SHAPES = [  # (n_img, H, W, Cin, Cout)
    (24, 64, 64, 320, 320),      # level 0: A 63 MB, W 1.8 MB
    (24, 64, 64, 320, 128),      # one column tile: A read once per patch (halo overlap only)
    (24, 32, 32, 640, 640),      # level 1: A 31 MB, W 7.4 MB
    (24, 16, 16, 2560, 1280),    # level 2 up path: A 31 MB, W 59 MB
    (6, 16, 16, 2560, 1280),     # same weights, a quarter of the rows: A 7.9 MB
    (6, 16, 16, 2560, 5120),     # W 236 MB
    (6, 16, 16, 2560, 7680),     # W 354 MB: beyond the Infinity Cache
]
if __name__ == "__main__":
    for (n, H, W, Cin, Cout) in SHAPES:
        x = torch.randn(n * H * W, Cin, device=dev, dtype=dt); w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(dt)
        b = torch.randn(Cout, device=dev)
        for _ in range(3): o.conv3x3(x, w, b, n, H, W)
        torch.cuda.synchronize()
        del x, w
