"""Tile-order sweep of the dense GEMM: emo_gemm_params.tile = tile | (gm << 4) pins the band height gm of the grouped order
(gemm_impl.h); 0 = the planner's choice."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from gemm_tiles import dense
for args, kw in (((98304, 2560, 320), dict(geglu=True, ln=True)), ((24576, 5120, 640), dict(geglu=True, ln=True)), ((6144, 10240, 1280), dict(geglu=True, ln=True)),
                 ((98304, 960, 320), dict(ln=True)), ((24576, 1920, 640), dict(ln=True)), ((6144, 3840, 1280), dict(ln=True))):
    dense(*args, (4, 4 | (1 << 4), 4 | (2 << 4), 4 | (4 << 4), 4 | (8 << 4)), **kw)
for args, kw in (((98304, 320, 1280), dict(res=True)), ((24576, 640, 2560), dict(res=True)), ((6144, 1280, 5120), dict(res=True)), ((24576, 640, 640), dict(res=True)),
                 ((6144, 1280, 1280), dict(res=True))):
    dense(*args, (0, 0 | (1 << 4), 0 | (2 << 4), 0 | (4 << 4), 0 | (8 << 4)), **kw)
