"""Tile x split-K sweep of the under-filled GEMMs (the 8x8 / 16x16 levels: M = 1536 / 6144) - run once per ring-depth variant
(EMO_HIP_LIB=emote_hack_amd/lib/variants/ns3.so ...)."""
import os
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from small_m_sweep_lib import conv, dense  # noqa: E402
print("lib:", os.environ.get("EMO_HIP_LIB", "product"), flush=True)
A0 = [(0, None)]
D = A0 + [(1, 1), (2, 1), (2, 2), (2, 4), (3, 1), (4, 1)]
dense(1536, 1280, 1280, D, res=True)
dense(1536, 3840, 1280, A0 + [(1, 1), (2, 1), (4, 1)], ln=True)
dense(1536, 1280, 5120, A0 + [(1, 1), (2, 1), (2, 2), (2, 4), (2, 8), (4, 1)], res=True)
dense(1536, 10240, 1280, A0 + [(2, 1), (4, 1)], geglu=True, ln=True)
dense(1536, 1280, 2560, A0 + [(1, 1), (2, 1), (2, 2), (2, 4)], res=True)
dense(6144, 1280, 1280, D, res=True)
dense(6144, 3840, 1280, A0 + [(2, 1), (4, 1)], ln=True)
dense(6144, 1280, 5120, A0 + [(2, 1), (2, 2), (4, 1)], res=True)
dense(6144, 1280, 2560, A0 + [(2, 1), (2, 2), (4, 1)], res=True)
dense(24576, 640, 640, A0 + [(2, 1), (3, 1)], res=True)
conv(24, 8, 8, 1280, 1280, A0 + [(2, 1), (2, 2), (2, 4), (2, 8), (4, 2), (4, 4), (4, 8)])
conv(24, 8, 8, 2560, 1280, A0 + [(2, 4), (2, 8), (4, 4), (4, 8)])
