import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(M,N,K):
    a = torch.randn(M,K,device=dev,dtype=dt); w = torch.randn(N,K,device=dev,dtype=dt)/30
    f = lambda: o.gemm(a,w,None)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/10*1e3
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
run(8192,8192,8192)
run(98304,2560,320)
run(24576,640,2560)
run(98304,320,320)
