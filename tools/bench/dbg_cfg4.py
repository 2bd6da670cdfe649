import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_fullsize import build, _run_loop
from tests import cases
from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
from emote_hack_amd import unet as U
torch.manual_seed(0)
unet = build(cases.SD15_MOTION, torch.bfloat16)
ref = build(cases.SD15, torch.bfloat16, cases.REF_PREFIX, cls=AppearanceEncoderModel)
for share in (True, False):
    U.SHARE_CFG_PREFIX = share
    _, e48 = _run_loop(unet, ref, 1, graphs=False, ref_group=2, frames=48)
    _, e12 = _run_loop(unet, ref, 1, graphs=False, ref_group=2, frames=48, frame_slice=slice(0, 12))
    _, e12b = _run_loop(unet, ref, 1, graphs=False, ref_group=2, frames=48, frame_slice=slice(0, 12))
    d = (e48[0][:, :, :12] - e12[0]).abs()
    print('share', share, 'equal', torch.equal(e48[0][:, :, :12], e12[0]), 'max', float(d.max()), 'frac nonzero', float((d > 0).float().mean()),
          'rerun equal', torch.equal(e12[0], e12b[0]), flush=True)
    for f in range(12):
        print('  frame', f, float(d[0, :, f].max()))
