"""SURVEY 8(e) on the GPU box: world_size 2 / 3 / 4 runs of the product loop with the real HIP kernels.  On a box with ONE MI355X
RCCL cannot run (it takes one device per rank): the ranks share cuda:0 and exchange through gloo (tests/dist_gpu_worker.py).  As
soon as MORE than one device is visible the second test runs the same loop over RCCL ("nccl"), one device per rank, at world size 2
(and 4 / 8 when that many GPUs are there) - eps all_gather per step and the ReferenceNet bank all_gather per group on the one
communicator, look-ahead stream on.  Targets: the reference-generated loop goldens."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, *args, backend="gloo", timeout=900):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    env["EMO_DIST_BACKEND"] = backend
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's hipIpc handles need it on this host driver
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), *map(str, args)],
                       capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout and f"backend={backend}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


# (world, golden prefix, graphs, reference_group, look-ahead stream, context_batch_size, guidance_scale)
@pytest.mark.parametrize("world,kind,graphs,ref_group,lookahead,cbs,gs", [
    (2, "ddpm", 1, 2, 1, 1, 7.5),        # 3 windows = 6 units over 2 ranks, two ReferenceNet groups, look-ahead next to the collectives
    (2, "ddim", 0, 10, 0, 1, 7.5),
    (3, "ddpm", 1, 1, 0, 1, 7.5),        # uneven deal, per-step ReferenceNet order, one timestep per group dealt over 3 ranks
    (4, "ddim", 1, 10, 0, 1, 7.5),       # more ranks than a rank's fair share: 6 units over 4 ranks (slots padded)
    (2, "ddim_cbs2", 1, 10, 0, 2, 7.5),  # literal context_batch_size > 1 pairing: two ReferenceNet variants exchanged
    (2, "ddim_nocfg", 1, 10, 0, 1, 1.0),
])
def test_multi_rank_loop_with_hip_kernels_matches_the_loop_goldens(world, kind, graphs, ref_group, lookahead, cbs, gs):
    _run(world, kind, graphs, ref_group, lookahead, cbs, gs)


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("world,kind,graphs,ref_group,lookahead,cbs,gs", [
    (2, "ddpm", 1, 2, 1, 1, 7.5),        # graphs + the look-ahead write pass next to both RCCL collectives, two ReferenceNet groups
    (2, "ddim", 0, 10, 0, 1, 7.5),       # eager launches, everything on the main stream
    (2, "ddim_cbs2", 1, 10, 1, 2, 7.5),  # two ReferenceNet variants exchanged
    (4, "ddim", 1, 10, 1, 1, 7.5),       # 6 units over 4 ranks (slots padded)
    (8, "ddpm", 1, 2, 1, 1, 7.5),        # more ranks than units: ranks 6, 7 idle but take part in every collective
])
def test_multi_rank_loop_over_rccl_one_device_per_rank(world, kind, graphs, ref_group, lookahead, cbs, gs):
    """The RCCL transport itself (torch.distributed backend "nccl", one MI355X per rank): skipped on the 1-GPU box, runs wherever
    `torch.cuda.device_count()` allows.  Same goldens, same bit-identical-latents-on-every-rank check as the gloo runs above."""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs for one device per rank (RCCL), {_n_gpus()} visible")
    _run(world, kind, graphs, ref_group, lookahead, cbs, gs, backend="nccl", timeout=600)


def test_two_ranks_at_full_size_equal_the_single_process_loop():
    """512x512 latents, two 12-frame windows over two ranks (sharing the one GPU over gloo; over RCCL with one device per rank when two
    are visible), f32, graphs + look-ahead + two ReferenceNet groups: the ranks hold bit-identical latents and they equal the single-process
    loop at the loop tolerance 2e-3 / 5e-4 (tests/dist_gpu_worker.py fullsize)."""
    _run(2, "fullsize", backend="nccl" if _n_gpus() >= 2 else "gloo", timeout=900)
