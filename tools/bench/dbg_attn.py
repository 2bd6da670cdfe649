import torch, sys
sys.path.insert(0, '.')
from emote_hack_amd import ops as o
from emote_hack_amd.synth import seeded_randn
DEV='cuda'
def attn_ref(qh, k, v, scale):
    s = torch.matmul(qh, k.transpose(-1, -2)) * scale
    return torch.matmul(s.softmax(-1), v)
def run(Bc, Fr, L, heads, d, Lb, first):
    C_ = heads*d; nb = Bc*Fr
    qq, kk, vv = (seeded_randn((nb, L, C_), s) for s in (27, 28, 29))
    bk, bv = seeded_randn((Bc, Lb, C_), 30), seeded_randn((Bc, Lb, C_), 31)
    sp = lambda t: t.reshape(t.shape[0], -1, heads, d).permute(0, 2, 1, 3)
    refs = []
    for b in range(nb):
        k_, v_ = kk[b:b+1], vv[b:b+1]
        if b >= first:
            k_ = torch.cat([k_, bk[b//Fr:b//Fr+1]], 1); v_ = torch.cat([v_, bv[b//Fr:b//Fr+1]], 1)
        refs.append(attn_ref(sp(qq[b:b+1]), sp(k_), sp(v_), d**-0.5).permute(0,2,1,3).reshape(L, C_))
    ref = torch.cat(refs)
    dv = lambda t: t.to(DEV)
    got = o.attention(dv(qq.reshape(-1, C_)), dv(kk.reshape(-1, C_)), dv(vv.permute(0,2,1).contiguous()), L, B=nb, Lq=L, heads=heads, d=d, scale=d**-0.5,
                      k1=dv(bk.reshape(-1, C_)), v1t=dv(bv.permute(0,2,1).contiguous()), Lk1=Lb, seg1_div=Fr, seg1_first_batch=first).cpu()
    err = (got-ref).abs().reshape(nb, L, heads, d)
    print(f"Bc={Bc} Fr={Fr} L={L} h={heads} d={d} Lb={Lb} first={first}: per-batch max err", [f"{float(err[b].max()):.2e}" for b in range(nb)],
          "per-head", [f"{float(err[:, :, h].max()):.2e}" for h in range(heads)])
run(2,3,48,4,40,80,3)
run(1,1,64,1,32,64,0)
run(1,1,64,1,32,32,0)
run(1,1,32,1,32,32,0)
run(2,1,64,1,32,64,0)
run(2,1,64,1,32,64,1)
run(1,2,64,2,32,64,0)
run(2,3,64,4,32,64,3)
