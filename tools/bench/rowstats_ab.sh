# LayerNorm statistics from the producing GEMM's epilogue: kernel + model tests, then the bench A/B inside one box
O=gpurun_out/${1:-r04p}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "row_statistics or layernorm_fold" 2>&1 | tail -8) > $O/k_tests.txt; cat $O/k_tests.txt
(timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8) > $O/m_tests.txt; cat $O/m_tests.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-profile --ln-stats-kernel > $O/b_ker_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-profile > $O/b_epi_$i.json 2>/dev/null
done
EMO_BENCH_SHAPES=$O/shapes.md python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
EMO_BENCH_SHAPES=$O/shapes_ker.md python bench.py --no-cpu-baseline --ln-stats-kernel > $O/bench_ker.json 2>>$O/bench.err
python - $O <<'PY'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+"/b_*.json"))+[O+"/bench.json",O+"/bench_ker.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
for n in ("bench.json","bench_ker.json"):
    d=json.loads(open(O+"/"+n).read().strip().splitlines()[-1])
    print(n, {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
tail -3 $O/bench.err
