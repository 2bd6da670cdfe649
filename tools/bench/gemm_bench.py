import sys, time, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(M,N,K,geglu=False,res=False,conv=None,T=False,bias=True):
    if conv:
        n,H,W,Cin = conv
        a = torch.randn(n*H*W, Cin, device=dev, dtype=dt); w = torch.randn(N, 9*Cin, device=dev, dtype=dt)/50
    else:
        a = torch.randn(M,K,device=dev,dtype=dt); w = torch.randn(N,K,device=dev,dtype=dt)/30
    b = torch.randn(N,device=dev) if bias else None
    no = N//2 if geglu else N
    r = torch.randn(M,no,device=dev,dtype=dt) if res else None
    def f():
        if conv: return o.conv3x3(a,w,b,conv[0],conv[1],conv[2],residual=r)
        if T: return o.gemm(a,w,None,transpose_rows=4096,transpose_ld=4096)
        return o.gemm(a,w,b,geglu=geglu,residual=r)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    it=20; e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/it*1e3
    fl = 2.0*M*N*(9*conv[3] if conv else K)
    by = 2.0*(M*(conv[3] if conv else K) + M*no*(2 if res else 1)) + 2.0*N*(9*conv[3] if conv else K)
    print(f"M={M:6d} N={N:5d} K={(9*conv[3] if conv else K):5d} {'geglu' if geglu else ''}{' res' if res else ''}{' conv' if conv else ''}{' T' if T else ''}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e3:7.1f} GB/s")
run(98304,320,320,res=True)
run(98304,320,320)
run(98304,960,320)
run(98304,2560,320,geglu=True)
run(98304,320,1280,res=True)
run(24576,640,640,res=True)
run(24576,5120,640,geglu=True)
run(24576,640,2560,res=True)
run(6144,1280,1280,res=True)
run(6144,10240,1280,geglu=True)
run(6144,1280,5120,res=True)
run(98304,320,0,conv=(24,64,64,320),res=True)
run(24576,640,0,conv=(24,32,32,640),res=True)
run(6144,1280,0,conv=(24,16,16,1280),res=True)
run(98304,320,320,T=True)
run(8192,8192,8192,bias=False)
