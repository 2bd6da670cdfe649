"""Per-part ablation of the dense GEMM (variants built from tools/bench/patches/gemm_ablation.patch with -DEMO_GEMM_ABL=n:
1 no LDS-DMA after a block's first stage, 2 no fragment reads after a stage's first k-step, 3 no epilogue, 4 = 1 + 2,
5 = 4 without the per-stage barrier): python tools/bench/gemm_abl.py product lib1.so ...   (results are wrong by design)"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "product":
            env["EMO_HIP_LIB"] = os.path.abspath(lib)
        print(f"=== {lib}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
    sys.exit(0)
sys.path.insert(0, ".")
sys.path.insert(0, "tools/bench")
from gemm_tiles import dense
dense(8192, 8192, 8192, (4,))
dense(98304, 2560, 320, (4,), geglu=True, ln=True)
dense(24576, 5120, 640, (4,), geglu=True, ln=True)
dense(6144, 10240, 1280, (4,), geglu=True, ln=True)
dense(98304, 960, 320, (4,), ln=True)
dense(98304, 320, 320, (3,), res=True)
dense(24576, 640, 640, (2,), res=True)
dense(6144, 1280, 1280, (2,), res=True)
dense(98304, 320, 1280, (3,), res=True)
dense(24576, 640, 2560, (2,), res=True)
dense(6144, 1280, 5120, (2,), res=True)
dense(6144, 3840, 1280, (4,), ln=True)
dense(24576, 1920, 640, (4,), ln=True)
