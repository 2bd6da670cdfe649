import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(M,N,K,res=True,T=False):
    a = torch.randn(M,K,device=dev,dtype=dt); w = torch.randn(N,K,device=dev,dtype=dt)/30
    b = torch.randn(N,device=dev)
    r = torch.randn(M,N,device=dev,dtype=dt) if res and not T else None
    def f():
        if T: return o.gemm(a,w,None,transpose_rows=4096,transpose_ld=4096)
        return o.gemm(a,w,b,residual=r)
    for _ in range(3): y=f()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/20*1e3
    print(f"M={M:6d} N={N:5d} K={K:5d}{' T' if T else ''}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
run(98304,320,320)
run(98304,320,320,res=False)
run(98304,960,320,res=False)
run(98304,640,320,res=False)
run(98304,320,1280)
run(24576,640,640)
run(24576,1920,640,res=False)
run(24576,1280,640,res=False)
run(6144,1280,1280)
run(6144,3840,1280,res=False)
run(8192,320,320)
run(2048,640,640)
run(98304,320,320,T=True)
run(24576,640,640,T=True)
