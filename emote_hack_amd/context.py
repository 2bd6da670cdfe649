"""Context-window scheduler (magicanimate/pipelines/context.py:12-49).  Integer, bit-exact with the
reference (goldens in tests/golden/ints.json)."""
from __future__ import annotations

import math
from typing import Callable, Optional


def ordered_halving(val: int) -> float:
    """context.py:12-17: bit-reversed 64-bit fraction."""
    return int(f"{val:064b}"[::-1], 2) / (1 << 64)


def uniform(step: int = ..., num_steps: Optional[int] = None, num_frames: int = ..., context_size: Optional[int] = None,
            context_stride: int = 3, context_overlap: int = 4, closed_loop: bool = True):
    """context.py:20-42.  Generator of frame-index lists; wrap-around modulo num_frames."""
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    context_stride = min(context_stride, int(math.ceil(math.log2(num_frames / context_size))) + 1)
    for k in range(context_stride):
        context_step = 1 << k
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(int(ordered_halving(step) * context_step) + pad,
                       num_frames + pad + (0 if closed_loop else -context_overlap),
                       (context_size * context_step - context_overlap)):
            yield [e % num_frames for e in range(j, j + context_size * context_step, context_step)]


def get_context_scheduler(name: str) -> Callable:
    if name == "uniform":
        return uniform
    raise ValueError(f"Unknown context_overlap policy {name}")
