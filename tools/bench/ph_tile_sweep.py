"""Which of the path's non-ping-pong GEMM shapes the phase loop (tile 7) now serves better than their planned tile (0)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from gemm_tiles import dense
T = (0, 7)
dense(98304, 320, 1600, T, res=True)      # fused FF tail, 64x64 level
dense(24576, 640, 3200, T, res=True)
dense(6144, 1280, 6400, T, res=True)
dense(98304, 320, 320, T, res=True)       # out-projections
dense(24576, 640, 640, T, res=True)
dense(6144, 1280, 1280, T, res=True)
dense(98304, 320, 320, T)                 # proj_in
dense(98304, 640, 320, T, ln=True)        # q|k
dense(24576, 1280, 640, T, ln=True)
dense(6144, 2560, 1280, T, ln=True)
dense(98304, 960, 320, T, ln=True)        # temporal q|k|v
dense(24576, 640, 640, T, ln=True)        # cross-attention q
dense(6144, 1280, 1280, T, ln=True)
dense(1536, 1280, 1280, T, res=True)
