import os, sys, torch
TILE = int(os.environ.get('CONV_TILE', '0')) or None   # emo_gemm_params.tile: 1 / 2 pin the 8 / 16-row patch, +4 keeps one launch at N % 128 == 64
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(n,H,W,Cin,N,nrot=4,up=False):
    # rotate over several input buffers so that A is not served from the infinity cache
    xs=[torch.randn(n*H*W, Cin, device=dev, dtype=dt) for _ in range(nrot)]
    w = torch.randn(N, 9*Cin, device=dev, dtype=dt)/50; b=torch.randn(N,device=dev)
    k = 4 if up else 1
    rs=[torch.randn(k*n*H*W, N, device=dev, dtype=dt) for _ in range(nrot)]
    def f(i): return o.conv3x3(xs[i%nrot],w,b,n,H,W,residual=None if up else rs[i%nrot],upsample2x=up,tile=TILE)
    for i in range(3): f(i)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(0)
        with torch.cuda.graph(g, stream=s):
            for i in range(12): f(i)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/12*1e3
    M=k*n*H*W
    tag = " up" if up else "   "
    print(f"conv{tag} M={M:6d} N={N:5d} Cin={Cin:5d}: {us:8.1f} us  {2.0*M*N*9*Cin/us/1e6:7.1f} TF/s", flush=True)
run(24,64,64,320,320)
run(24,64,64,640,320)
run(24,32,32,640,640)
run(24,32,32,1280,640)
run(24,16,16,1280,1280)
run(24,16,16,2560,1280)
run(24,64,64,960,320)
run(24,32,32,640,640,up=True)
run(24,16,16,1280,1280,up=True)
run(24,8,8,1280,1280,up=True)
run(24,8,8,1280,1280)
run(24,8,8,2560,1280)
