import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(M,N,K,geglu=False,res=False):
    a = torch.randn(M,K,device=dev,dtype=dt); w = torch.randn(N,K,device=dev,dtype=dt)/30
    b = torch.randn(N,device=dev)
    no = N//2 if geglu else N
    r = torch.randn(M,no,device=dev,dtype=dt) if res else None
    f = lambda: o.gemm(a,w,b,geglu=geglu,residual=r)
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/10*1e3
    print(f"M={M:6d} N={N:5d} K={K:5d}{' geglu' if geglu else ''}{' res' if res else ''}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
run(98304,2560,320,geglu=True)
run(24576,5120,640,geglu=True)
run(6144,10240,1280,geglu=True)
run(98304,960,320)
run(24576,1920,640)
run(6144,3840,1280)
run(98304,320,1280,res=True)
run(24576,640,2560,res=True)
run(6144,1280,5120,res=True)
run(98304,640,320)
run(24576,1280,640)
