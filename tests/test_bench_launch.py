"""bench.py launch geometry (CPU): `python bench.py --gpus N` without a launcher must start N ranks ITSELF
(one process per GPU like the reference's `mp.spawn`, magicanimate/pipelines/animation.py:246-269) - the driver's plain
`python3 bench.py --gpus 8` used to benchmark one GPU silently."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_self_spawns_n_ranks_without_a_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"spawned_ranks": 2, "world_size": 2}


def test_bench_under_a_launcher_does_not_respawn():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn-check"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["spawned_ranks"] == 1
    # a launcher's world size that contradicts --gpus is an error, not a silent 1-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--spawn-check"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_spawn_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.spawn_command(["--gpus", "8", "--steps", "5", "--mode", "strong"], 8, port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--mode", "strong"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")
