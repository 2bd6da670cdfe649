"""PMC subject: the 8192^3 and one mid-size GEMM on emo_gemm (planned tile, staged 8-wave tile) and on hipBLASLt - run under
rocprofv3 --pmc (tools/bench/gemm_pmc2.sh) to compare wave-cycle breakdowns kernel by kernel."""
import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev = 'cuda'; dt = torch.bfloat16
def run(M, N, K):
    a = torch.randn(M, K, device=dev, dtype=dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=dt)
    for t in (0, 8):
        for _ in range(3): o.gemm(a, w, b, out=out, tile=t, split_k=1)
    b16 = b.to(dt)
    for _ in range(3): torch.addmm(b16, a, w.t(), out=out)
    torch.cuda.synchronize()
import os
run(*[int(x) for x in os.environ.get('GEMM_PMC_SHAPE', '8192,8192,8192').split(',')])
