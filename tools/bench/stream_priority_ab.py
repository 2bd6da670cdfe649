"""A/B: the sampling loop on a HIGH-priority stream (the ReferenceNet look-ahead stream keeps the default priority, i.e. it only gets what the
main stream leaves) against the default (both at the same priority).  Same process, alternating; ms per step over 50-step clips."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from emote_hack_amd import DDPMScheduler
from emote_hack_amd.pipeline import EMOAnimationPipeline
from emote_hack_amd.synth import seeded_randn

dev = torch.device('cuda', 0)
unet, ref = bench.build_models(dev, torch.bfloat16)
print('priority range', torch.cuda.Stream.priority_range(), flush=True)


def run(stream):
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDPMScheduler())
    with torch.cuda.stream(stream):
        st = pipe.prepare_denoise(seeded_randn((1, 4, 12, 64, 64), 1).to(dev), seeded_randn((1, 4, 64, 64), 3), seeded_randn((2, 77, 768), 2), appearance_encoder=ref,
                                  num_inference_steps=50, guidance_scale=7.5, context_frames=12, context_stride=1, context_overlap=0, seed=0, use_graphs=True, reference_group=10)
        for si in range(10):
            pipe.denoise_step(st, si)
        torch.cuda.synchronize()
        out = []
        for rep in range(3):
            t0 = time.perf_counter()
            for si in range(50):
                pipe.denoise_step(st, (10 + si) % 50)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 50 * 1e3)
    return out


for name, mk in (('default', lambda: torch.cuda.current_stream()), ('high-priority main', lambda: torch.cuda.Stream(priority=-1)),
                 ('default', lambda: torch.cuda.current_stream()), ('high-priority main', lambda: torch.cuda.Stream(priority=-1))):
    print(f'{name:20s}', ' '.join(f'{x:.3f}' for x in run(mk())), flush=True)
