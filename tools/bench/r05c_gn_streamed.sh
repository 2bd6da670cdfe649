# round 5: streamed GroupNorm (3 launches) vs the stats + apply pair - kernel tests, isolated timings, same-box A/B of the loop
O=gpurun_out/${1:-r05c}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "groupnorm" 2>&1 | tail -6) > $O/tests.txt; cat $O/tests.txt
python tools/bench/norm_bench.py 2>/dev/null | grep GN > $O/norm_streamed.txt
python - <<'PY' 2>/dev/null | grep GN > $O/norm_pair.txt
import runpy, sys
sys.path.insert(0, '.')
from emote_hack_amd import ops
ops.GN_STREAMED = False
runpy.run_path('tools/bench/norm_bench.py')
PY
paste -d'\n' $O/norm_streamed.txt $O/norm_pair.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-profile > $O/b_streamed_$i.json 2>$O/err_s_$i.txt
python bench.py --no-cpu-baseline --no-profile --gn-pair > $O/b_pair_$i.json 2>$O/err_p_$i.txt
done
EMO_BENCH_SHAPES=$O/shapes.md python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - $O <<'PY'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+"/b_*.json"))+[O+"/bench.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3))
    except Exception as e: print(f, "ERR", e)
d=json.loads(open(O+"/bench.json").read().strip().splitlines()[-1])
print({k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
grep groupnorm $O/shapes.md
