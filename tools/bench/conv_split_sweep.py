"""Split-K x tile sweep of the 8x8-level 3x3 convs (M = 1536: 120 tiles of 128x128 on 256 CUs)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from small_m_sweep_lib import conv, dense  # noqa: E402
for cin in (1280, 2560):
    conv(24, 8, 8, cin, 1280, [(0, None)] + [(t, s) for t in (2, 3, 4) for s in (4, 6, 8, 12, 16, 24)])
conv(10, 8, 8, 1280, 1280, [(0, None)] + [(t, s) for t in (2, 4) for s in (4, 8, 16, 32)])   # ReferenceNet group (T = 10)
conv(12, 8, 8, 1280, 1280, [(0, None)] + [(t, s) for t in (2, 4) for s in (4, 8, 16, 32)])   # shared-prefix / single-branch call
dense(1536, 1280, 5120, [(0, None)] + [(t, s) for t in (2, 4) for s in (2, 4, 8, 16)], res=True)
dense(1536, 1280, 2560, [(0, None), (1, 1)] + [(t, s) for t in (2, 4) for s in (2, 4, 8)], res=True)
