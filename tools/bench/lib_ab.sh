# product library vs a variant .so on the whole step, alternating inside one box:  lib_ab.sh OUTDIR VARIANT.so [pytest -k expr]
O=gpurun_out/$1; V=$PWD/$2; mkdir -p $O
if [ -n "$3" ]; then (timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "$3" 2>&1 | tail -5) > $O/tests.txt; cat $O/tests.txt; fi
for i in 1 2 3; do
EMO_HIP_LIB=$V python bench.py --no-cpu-baseline --no-profile > $O/b_var_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-profile > $O/b_prod_$i.json 2>/dev/null
done
python - $O <<'PY'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+"/b_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3))
PY
