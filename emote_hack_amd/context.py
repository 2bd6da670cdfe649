"""Context windows of the sampling loop: which frame indices every UNet call covers (magicanimate/pipelines/context.py:12-49).
INTEGER work, bit-exact with the reference: tests/golden/ints.json holds its outputs for eight (frames, window, stride, overlap)
geometries and four `step` values.

The schedule: for dilation 1, 2, 4, ... (at most `context_stride` levels, and never more than it takes one dilated window to span the clip)
windows of `context_size` frames, `dilation` apart inside a window, start every `context_size * dilation - context_overlap` frames and wrap
around the clip; `step` rotates the whole pattern by a low-discrepancy fraction of the clip (the pipeline always passes step 0,
EMOAnimationPipeline.py:718-720,748-750)."""
from __future__ import annotations

import math
from typing import Callable, Iterator, List, Optional

_TWO_64 = 1 << 64


def ordered_halving(val: int) -> float:
    """Radical inverse of a 64-bit integer: bit i of `val` lands on bit 63 - i, the result read as a fraction of 2^64 (context.py:12-17;
    the same int / int division, so the same double)."""
    rev = 0
    for i in range(64):
        rev = (rev << 1) | ((val >> i) & 1)
    return rev / _TWO_64


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True) -> Iterator[List[int]]:
    """context.py:20-42.  Yields the windows as lists of frame indices (modulo num_frames)."""
    n, size = num_frames, context_size
    if n <= size:                                   # the clip fits one window
        yield [*range(n)]
        return
    levels = min(context_stride, int(math.ceil(math.log2(n / size))) + 1)
    phase = ordered_halving(step)
    shift = int(round(n * phase))                   # rotation of the pattern, in frames
    stop = n + shift - (0 if closed_loop else context_overlap)
    for level in range(levels):
        dil = 1 << level
        start, hop = int(phase * dil) + shift, size * dil - context_overlap
        if hop == 0:
            raise ValueError("range() arg 3 must not be zero")     # what the reference's range(start, stop, 0) raises
        while hop > 0 and start < stop:
            yield [(start + i * dil) % n for i in range(size)]
            start += hop


def get_context_scheduler(name: str) -> Callable:
    if name != "uniform":
        raise ValueError(f"Unknown context_overlap policy {name}")
    return uniform
