#!/usr/bin/env python3
"""Where do the wave cycles of the dense GEMM go?  Needs the timing build
(PATCH=tools/bench/patches/gemm_timing.patch tools/bench/build_variant.sh timing; run with
EMO_HIP_LIB=emote_hack_amd/lib/variants/timing.so): every wave stamps s_memtime at four points of each K stage and sums
  A  wait for the stage's data (s_waitcnt before the barrier)      B  s_barrier
  C  first fragment read (+ the LDS-DMA issue of the next stage)     D  the MFMA k-steps
  E1 barrier in front of the epilogue   E2 the epilogue   E3 next tile's set-up + accumulator init (bias / LN loads)
Prints the per-stage averages over all waves and the share of each part."""
import sys

import torch

sys.path.insert(0, ".")
from emote_hack_amd import ops as o  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def run(M, N, K, tile, geglu=False, per_wave=False):
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    ws = torch.zeros(2048 * 8 * 12, device=dev, dtype=torch.int32)
    lib = o._lib.load()
    real = lib.emo_gemm

    def with_ws(pref, stream):   # the timing build writes its per-wave sums through emo_gemm_params.workspace
        pref._obj.workspace = ws.data_ptr()
        return real(pref, stream)
    lib.emo_gemm = with_ws
    try:
        for _ in range(2):
            ws.zero_()
            o.gemm(a, w, b, tile=tile, split_k=1, geglu=geglu)
        torch.cuda.synchronize()
    finally:
        lib.emo_gemm = real
    t = ws.view(-1, 12).cpu().long()
    t = t[t[:, 9] == 1]
    n = t[:, 8].float().mean()
    tot = t[:, 7].float().mean()
    parts = [t[:, i].float().mean() for i in range(7)]
    names = "A:data-wait B:barrier C:first-read(+issue) D:k-steps E1:epi-barrier E2:epilogue E3:next-tile-init".split()
    s = f"M={M} N={N} K={K} tile {tile}{' geglu' if geglu else ''}: {len(t)} waves, {n:.0f} stages/wave, {tot:.0f} cycles/wave |"
    for nm, v in zip(names, parts):
        s += f" {nm} {v / n:6.0f}/stage {100 * v / tot:4.1f}% |"
    s += f" other {100 * (tot - sum(parts)) / tot:4.1f}%"
    print(s, flush=True)
    if per_wave:   # means per wave slot of the block (waves w and w + 4 share a SIMD): who waits at the barrier?
        nw = 8 if (tile & 15) in (0, 4, 8) and N >= 1792 else 4
        v = t[: len(t) // nw * nw].view(-1, nw, 12).float().mean(0)
        for wv in range(nw):
            print(f"    wave {wv}: " + " ".join(f"{nm.split(':')[0]} {v[wv, i] / v[wv, 8]:6.0f}" for i, nm in enumerate(names)), flush=True)


if __name__ == "__main__":
    for tile in (0, 8):
        run(8192, 8192, 8192, tile)
        run(24576, 5120, 640, tile)
        run(98304, 2560, 320, tile, geglu=True)
        run(6144, 10240, 1280, tile, geglu=True)
    for tile in (0, 10, 9):
        run(98304, 320, 320, tile)
        run(24576, 640, 640, tile)
        run(6144, 1280, 1280, tile)
