"""round 5: is the planner's tile still the best one on the residual / tail / LN projections after the loader change?
(tile ids: 0 planned, 1 64x64, 2 128x128, 3 128x160, 4 256x256, 7 256x256 phase loop)"""
import runpy, sys
sys.argv = ["gemm_tiles.py"]
m = runpy.run_path("tools/bench/gemm_tiles.py", run_name="sweep")
dense = m["dense"]
T = (0, 1, 2, 3, 4, 7)
for args, kw in (((98304, 320, 320), dict(res=True)), ((98304, 320, 1600), dict(res=True)), ((98304, 320, 320), dict(ln=True)),
                 ((24576, 640, 640), dict(res=True)), ((24576, 640, 3200), dict(res=True)), ((24576, 640, 640), dict(ln=True)),
                 ((6144, 1280, 1280), dict(res=True)), ((6144, 1280, 6400), dict(res=True)), ((6144, 1280, 1280), dict(ln=True)),
                 ((1536, 1280, 1280), dict(res=True)), ((49152, 320, 320), dict(res=True))):
    dense(*args, T, **kw)
