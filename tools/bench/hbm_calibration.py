"""What HBM rate does THIS box give plain streaming kernels?  Calibrates the 'HBM-bound' statements of DESIGN.md (peak 8 TB/s per the guide):
torch device-to-device copies (read + write) and the product's read-only LayerNorm-statistics pass, at one level-0 activation (63 MB), at 1 GB
(beyond the 256 MB Infinity Cache) and rotating over buffers so that nothing is served from cache."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'


def timeit(f, n=20):
    for i in range(3): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb, nrot in ((63, 16), (252, 8), (1008, 4)):
    n = mb * 1024 * 1024 // 2
    xs = [torch.randn(n // 320, 320, device=dev, dtype=torch.bfloat16) for _ in range(nrot)]
    ys = [torch.empty_like(x) for x in xs]
    t = timeit(lambda i: ys[i % nrot].copy_(xs[i % nrot]))
    b = xs[0].numel() * 2
    ts = timeit(lambda i: o.layer_norm_stats(xs[i % nrot]))
    tl = timeit(lambda i: o.layer_norm(xs[i % nrot], torch.ones(320, device=dev), torch.zeros(320, device=dev)))
    print(f"{mb:5d} MB x{nrot}: torch copy {2 * b / t / 1e12:5.2f} TB/s (r+w) | emo_layernorm_stats {b / ts / 1e12:5.2f} TB/s (read) | emo_layernorm {2 * b / tl / 1e12:5.2f} TB/s (r+w)", flush=True)
