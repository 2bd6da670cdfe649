"""Build the gfx950 kernel library in-tree: hipcc --offload-arch=gfx950 -> emote_hack_amd/lib/libemo_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so travels to the
GPU box with the repo snapshot (git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libemo_hip.so")
# the GEMM and the halo-conv kernels are instantiated per element type in their own translation units (parallel compile)
SOURCES = ["elementwise.hip", "norm.hip", "gemm.hip", "gemm_f32.hip", "gemm_bf16.hip", "gemm_f16.hip", "conv_halo_f32.hip", "conv_halo_bf16.hip",
           "conv_halo_f16.hip", "attention.hip", "temporal.hip", "conditioning.hip", "frontend.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-pass-failed"]


def csrc_digest() -> str:
    """sha256 over the kernel sources + the C header: names the BUILD a profile was taken on (tools/pmc_to_json.py stamps it into
    profiles/r*_pmc.json / r*_mfma.json; bench.py attaches counter figures only from a profile of the running build)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in files + [os.path.join(os.path.dirname(HERE), "include", "emo_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build_extension(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_api.h"), os.path.join(CSRC, "gemm_impl.h"), os.path.join(CSRC, "conv_halo_impl.h"),
            os.path.join(os.path.dirname(HERE), "include", "emo_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])
    if jobs or not os.path.exists(LIB):
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
            list(ex.map(run, jobs))
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
        # every kernel stub must resolve NOW (RTLD_NOW): a kernel template the host pass dropped links fine and fails at load time
        import ctypes
        ctypes.CDLL(LIB, mode=getattr(os, "RTLD_NOW", 2))
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv))
