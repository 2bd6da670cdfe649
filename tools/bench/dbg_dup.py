import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_fullsize import build
from tests import cases
from emote_hack_amd.synth import seeded_randn
torch.manual_seed(0)
unet = build(cases.SD15_MOTION, torch.bfloat16)
x = seeded_randn((1, 4, 12, 64, 64), 1).repeat(2, 1, 1, 1, 1).cuda()
ctx = seeded_randn((2, 77, 768), 2).cuda()
for trial in range(2):
    junk = [torch.full((50_000_000 * (trial + 1),), float('nan'), device='cuda', dtype=torch.bfloat16) for _ in range(3)]
    del junk
    y0 = unet(x, 981, ctx).sample.float()
    junk = [torch.full((30_000_000 * (trial + 1),), float('nan'), device='cuda', dtype=torch.bfloat16) for _ in range(3)]
    del junk
    y1 = unet(x, 981, ctx, _halves_identical=True).sample.float()
    d = (y0 - y1).abs()
    print('trial', trial, 'nan', int(torch.isnan(y1).sum()), 'max', float(d.nan_to_num(9e9).max()), 'mean', float(d.nan_to_num(0).mean()), 'ref mean', float(y0.abs().mean()), flush=True)
