# GroupNorm one-launch A/B inside one box: kernel tests, then the bench with / without the one-launch kernel
O=gpurun_out/${1:-r04n}; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "groupnorm" 2>&1 | tail -5) > $O/gn_tests.txt; cat $O/gn_tests.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-profile --gn-two-launch > $O/b_two_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-profile > $O/b_one_$i.json 2>/dev/null
done
EMO_BENCH_SHAPES=$O/shapes.md python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
EMO_BENCH_SHAPES=$O/shapes_two.md python bench.py --no-cpu-baseline --gn-two-launch > $O/bench_two.json 2>>$O/bench.err
python - $O <<'PY'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+"/b_*.json"))+[O+"/bench.json",O+"/bench_two.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"])
d=json.loads(open(O+"/bench.json").read().strip().splitlines()[-1])
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3))
PY
echo one; grep "groupnorm " $O/shapes.md; echo two; grep "groupnorm " $O/shapes_two.md
