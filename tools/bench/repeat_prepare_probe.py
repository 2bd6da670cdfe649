"""Probe: does a SECOND / THIRD prepare_denoise + loop in one process run as fast as the first?  (stream_priority_ab.py saw one 57 ms run.)"""
import gc, sys, time, torch
sys.path.insert(0, '.')
import bench
from emote_hack_amd import DDPMScheduler
from emote_hack_amd.pipeline import EMOAnimationPipeline
from emote_hack_amd.synth import seeded_randn

dev = torch.device('cuda', 0)
unet, ref = bench.build_models(dev, torch.bfloat16)


def run(stream, cleanup):
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDPMScheduler())
    with torch.cuda.stream(stream):
        st = pipe.prepare_denoise(seeded_randn((1, 4, 12, 64, 64), 1).to(dev), seeded_randn((1, 4, 64, 64), 3), seeded_randn((2, 77, 768), 2), appearance_encoder=ref,
                                  num_inference_steps=50, guidance_scale=7.5, context_frames=12, context_stride=1, context_overlap=0, seed=0, use_graphs=True, reference_group=10)
        for si in range(10):
            pipe.denoise_step(st, si)
        torch.cuda.synchronize()
        out = []
        for rep in range(2):
            t0 = time.perf_counter()
            for si in range(50):
                pipe.denoise_step(st, (10 + si) % 50)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 50 * 1e3)
    mem = torch.cuda.memory_reserved() / 2**30
    if cleanup:
        del st, pipe
        gc.collect()
        torch.cuda.empty_cache()
    return out, mem


seq = sys.argv[1] if len(sys.argv) > 1 else "DDDHDH"
cleanup = len(sys.argv) > 2 and sys.argv[2] == "clean"
for ch in seq:
    s = torch.cuda.current_stream() if ch == "D" else torch.cuda.Stream(priority=-1 if ch == "H" else 0)
    o, mem = run(s, cleanup)
    print(ch, ' '.join(f'{x:.3f}' for x in o), f'reserved {mem:.1f} GiB', flush=True)
