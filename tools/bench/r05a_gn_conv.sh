# round 5, call 1: GroupNorm + SiLU inside the halo conv - kernel tests, UNet goldens, then same-box A/B of the loop:
#   new build fused / new build unfused (--gn-conv-min-hw 0: the buffer-descriptor loaders alone).  The round-4 legs of profiles/r05b_gn_conv.txt ran the same
#   bench from a scratch checkout of the round-4 commit (git archive 841cd58 -> tools/bench/_r04_tree, its library built from that tree) inside the same call;
#   att_r04.so = this build linked with the round-4 attention.o (tools/bench/attn_bench.py A/B)
O=gpurun_out/${1:-r05a}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3 or groupnorm" 2>&1 | tail -8) > $O/tests_kernels.txt; cat $O/tests_kernels.txt
(timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -8) > $O/tests_attention.txt; cat $O/tests_attention.txt
python tools/bench/attn_bench.py product emote_hack_amd/lib/variants/att_r04.so > $O/attention_ab.txt 2>&1; cat $O/attention_ab.txt
(timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_controlnet.py -m gpu -x -q 2>&1 | tail -8) > $O/tests_unet.txt; cat $O/tests_unet.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-profile > $O/b_fused_$i.json 2>$O/err_fused_$i.txt
python bench.py --no-cpu-baseline --no-profile --gn-conv-min-hw 0 > $O/b_unfused_$i.json 2>$O/err_unfused_$i.txt
python bench.py --no-cpu-baseline --no-profile --gn-conv-min-hw 1024 > $O/b_fused1024_$i.json 2>$O/err_f1024_$i.txt

done
EMO_BENCH_SHAPES=$O/shapes.md python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
EMO_BENCH_SHAPES=$O/shapes_unfused.md python bench.py --no-cpu-baseline --gn-conv-min-hw 0 > $O/bench_unfused.json 2>>$O/bench.err

python - $O <<'PY'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+"/b_*.json"))+[O+"/bench.json",O+"/bench_unfused.json",O+"/bench_r04.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3))
    except Exception as e: print(f, "ERR", e)
for n in ("bench","bench_unfused","bench_r04"):
    try:
        d=json.loads(open(O+f"/{n}.json").read().strip().splitlines()[-1])
        print(n, {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
    except Exception as e: print(n, "ERR", e)
PY
