"""Worker of tests/test_gpu_dist.py: one rank of a world_size > 1 run of the product sampling loop with the REAL HIP kernels.
RCCL wants one device per rank and the GPU box has one, so the ranks share cuda:0 and talk through gloo (device tensors are
staged by the backend): what is exercised is everything but the transport - unit dealing, the bank exchange and its re-ordering,
the eps exchange, redundant accumulate + sampler step, HIP graphs and the look-ahead stream next to collectives."""
import os
import sys

import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kind, graphs, ref_group, lookahead, cbs, gs = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3]), sys.argv[4] == "1", int(sys.argv[5]), float(sys.argv[6])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("EMO_DIST_BACKEND", "gloo") == "nccl":    # one device per rank: the RCCL transport itself (>= `world` GPUs visible)
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        td.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
    else:
        td.init_process_group("gloo")
    from safetensors.torch import load_file
    from emote_hack_amd import DDIMScheduler, DDPMScheduler
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    from emote_hack_amd.synth import seeded_randn
    from tests import cases
    from tests.test_gpu_unet import build
    g = load_file(os.path.join(cases.GOLDEN_DIR, "loop_tiny.safetensors"))
    ref = build(cases.TINY, torch.float32, cases.REF_PREFIX, cls=AppearanceEncoderModel, has_out=False)
    unet = build(cases.TINY_MOTION, torch.float32)
    pipe = EMOAnimationPipeline(unet=unet, scheduler=DDIMScheduler() if kind.startswith("ddim") else DDPMScheduler())
    text = seeded_randn((2, 5, 32), 2)
    lat = pipe.denoise(seeded_randn((1, 4, 8, 16, 16), 5).to("cuda"), seeded_randn((1, 4, 16, 16), 3), text if gs > 1 else text[1:],
                       appearance_encoder=ref, num_inference_steps=3, guidance_scale=gs, context_frames=4, context_stride=1,
                       context_overlap=2, seed=0, use_graphs=graphs, dist=True, rank=rank, world_size=world, reference_group=ref_group,
                       reference_lookahead=lookahead, context_batch_size=cbs)
    torch.cuda.synchronize()
    torch.testing.assert_close(lat.cpu(), g[f"{kind}/latents"], rtol=1e-3, atol=1e-4)
    # every rank holds bit-identical latents without a broadcast
    all_l = torch.zeros(world, lat.numel(), device="cuda")
    td.all_gather_into_tensor(all_l.view(-1), lat.reshape(-1).contiguous())
    for r in range(world):
        assert torch.equal(all_l[r], all_l[0]), f"rank {r} differs"
    if rank == 0:
        print(f"DIST_GPU_OK world={world} backend={td.get_backend()} {kind} graphs={graphs} ref_group={ref_group} lookahead={lookahead}", flush=True)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
