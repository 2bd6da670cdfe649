import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(M,N,K,conv=None,res=True,check=False):
    if conv:
        n,H,W,Cin = conv
        a = torch.randn(n*H*W, Cin, device=dev, dtype=dt); w = torch.randn(N, 9*Cin, device=dev, dtype=dt)/50
    else:
        a = torch.randn(M,K,device=dev,dtype=dt); w = torch.randn(N,K,device=dev,dtype=dt)/30
    b = torch.randn(N,device=dev)
    r = torch.randn(M,N,device=dev,dtype=dt) if res else None
    def f():
        if conv: return o.conv3x3(a,w,b,conv[0],conv[1],conv[2],residual=r)
        return o.gemm(a,w,b,residual=r)
    for _ in range(3): y=f()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/20*1e3
    KK = 9*conv[3] if conv else K
    fl = 2.0*M*N*KK
    err = ''
    if check and not conv:
        ref = (a.float()@w.float().t() + b + (r.float() if res else 0))
        err = f" maxerr {(y.float()-ref).abs().max().item():.3f}"
    print(f"M={M:6d} N={N:5d} K={KK:5d}{' conv' if conv else ''}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s{err}", flush=True)
run(1536,1280,1280,check=True)
run(1536,1280,5120)
run(1536,1280,0,conv=(24,8,8,1280))
run(1536,1280,0,conv=(24,8,8,2560))
run(6144,1280,1280)
run(6144,640,640)
run(8192,320,320,check=True)
run(2048,640,640)
run(512,1280,1280)
run(128,1280,1280)
run(8192,320,0,conv=(2,64,64,320))
run(2048,640,0,conv=(2,32,32,640))
run(512,1280,0,conv=(2,16,16,1280))
run(128,1280,0,conv=(2,8,8,1280))
run(6144,1280,0,conv=(24,16,16,1280))
