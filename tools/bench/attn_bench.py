import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(B,L,Lk1,heads,d):
    C=heads*d
    q=torch.randn(B*L,C,device=dev,dtype=dt); k=torch.randn(B*L,C,device=dev,dtype=dt); vt=torch.randn(B,C,L,device=dev,dtype=dt)
    kw={}
    if Lk1:
        kw=dict(k1=torch.randn(2*Lk1,C,device=dev,dtype=dt), v1t=torch.randn(2,C,Lk1,device=dev,dtype=dt), Lk1=Lk1, seg1_div=B//2, seg1_first_batch=0)
    f=lambda: o.attention(q,k,vt,L,B=B,Lq=L,heads=heads,d=d,scale=d**-0.5,**kw)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/10*1e3
    fl=4.0*B*heads*L*(L+Lk1)*d
    print(f"B={B} L={L}+{Lk1} h={heads} d={d}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s")
run(24,4096,4096,8,40); run(24,4096,0,8,40); run(2,4096,0,8,40); run(24,1024,1024,8,80); run(24,1024,0,8,80); run(24,256,256,8,160); run(24,64,64,8,160)
