import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(B,F,HW,heads,d):
    C=heads*d
    qkv=torch.randn(B*F*HW,3*C,device=dev,dtype=dt)
    f=lambda:o.temporal_attention(qkv,B,F,HW,heads,d,d**-0.5)
    for _ in range(3): f()
    g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g,stream=s):
            for _ in range(10): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/10*1e3
    by=B*F*HW*C*2*4
    # reference check on a slice
    x=qkv.float().view(B,F,HW,3,heads,d)[:, :, :64]
    q,k,v=x[...,0,:,:],x[...,1,:,:],x[...,2,:,:]
    sc=torch.einsum('bfphd,bgphd->bphfg',q,k)*d**-0.5
    ref=torch.einsum('bphfg,bgphd->bfphd',sc.softmax(-1),v).reshape(B,F,64,C)
    got=f().float().view(B,F,HW,C)[:, :, :64]
    print(f"temporal B={B} F={F} HW={HW} C={C}: {us:7.1f} us  {by/us/1e3:6.0f} GB/s  maxerr {(got-ref).abs().max().item():.4f}", flush=True)
run(2,12,4096,8,40)
run(2,12,1024,8,80)
run(2,12,256,8,160)
run(2,12,64,8,160)
