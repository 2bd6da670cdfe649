import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def timeit(f, n=20):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
def gn(M,C,n_inst):
    x=torch.randn(M,C,device=dev,dtype=dt); g=torch.randn(C,device=dev); b=torch.randn(C,device=dev)
    us=timeit(lambda:o.group_norm(x,g,b,n_inst,32,1e-5,True))
    y=o.group_norm(x,g,b,n_inst,32,1e-5,True).float()
    ref=torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().view(n_inst,M//n_inst,C).permute(0,2,1),32,g,b,1e-5)).permute(0,2,1).reshape(M,C)
    print(f"GN M={M:6d} C={C:5d} inst={n_inst:3d}: {us:7.1f} us  {M*C*2*3/us/1e3:7.0f} GB/s(3 pass)  err {(y-ref).abs().max().item():.3f}", flush=True)
def ln(M,C):
    x=torch.randn(M,C,device=dev,dtype=dt); g=torch.randn(C,device=dev); b=torch.randn(C,device=dev)
    us=timeit(lambda:o.layer_norm(x,g,b))
    print(f"LN M={M:6d} C={C:5d}: {us:7.1f} us  {M*C*2*2/us/1e3:7.0f} GB/s", flush=True)
for M,C in ((98304,320),(24576,640),(6144,1280),(1536,1280),(24576,320),(24576,960),(6144,1920),(1536,2560),(98304,640)):
    gn(M,C,2); 
    if C in (320,640,1280): gn(M,C,24)
for M,C in ((8192,320),(2048,640),(512,1280),(128,1280)):
    gn(M,C,2)
for M,C in ((98304,320),(24576,640),(6144,1280),(1536,1280),(8192,320),(2048,640),(512,1280)):
    ln(M,C)
