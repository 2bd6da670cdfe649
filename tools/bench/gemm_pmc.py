import sys, torch
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
dev='cuda'; dt=torch.bfloat16
def run(M,N,K,geglu=False,res=False):
    a = torch.randn(M,K,device=dev,dtype=dt); w = torch.randn(N,K,device=dev,dtype=dt)/30
    b = torch.randn(N,device=dev); no = N//2 if geglu else N
    r = torch.randn(M,no,device=dev,dtype=dt) if res else None
    for _ in range(3): o.gemm(a,w,b,geglu=geglu,residual=r)
    torch.cuda.synchronize()
run(98304,320,320,res=True)
run(98304,2560,320,geglu=True)
run(24576,640,640,res=True)
run(6144,1280,1280,res=True)
run(6144,1280,5120,res=True)
run(8192,8192,8192)
