"""Lockstep (tile 4) vs ping-pong (tile 7) main loop of the 256x256 dense tile on the path's shapes and 8192^3."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools/bench")
from gemm_tiles import dense  # noqa: E402
T = (0, 4, 7)
dense(8192, 8192, 8192, T)
dense(98304, 2560, 320, T, geglu=True, ln=True)
dense(24576, 5120, 640, T, geglu=True, ln=True)
dense(6144, 10240, 1280, T, geglu=True, ln=True)
dense(98304, 960, 320, T, ln=True)
dense(24576, 1920, 640, T, ln=True)
dense(6144, 3840, 1280, T, ln=True)
dense(98304, 320, 1600, T, res=True)
dense(24576, 640, 3200, T, res=True)
dense(6144, 1280, 6400, T, res=True)
dense(24576, 1280, 640, T, ln=True)
dense(6144, 1280, 1280, T, res=True)
dense(24576, 640, 640, T, res=True)
