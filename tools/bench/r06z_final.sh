#!/bin/bash
# round-6 final measurement set on ONE box (from the repo root on the GPU box): profile passes of the default bench command, the
# bench line itself (with cpu_baseline + parity), per-shape table, and the other entry points / configurations.
set -u
OUT=$PWD/gpurun_out/r06z
mkdir -p $OUT
tools/profile_round.sh r06z > $OUT/profile_round.log 2>&1
EMO_BENCH_SHAPES=$OUT/shapes.md python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-profile --clips 0 > $OUT/bench_50steps.json 2>> $OUT/bench.err
python bench.py --entry call --no-cpu-baseline > $OUT/bench_call.json 2>> $OUT/bench.err
EMO_BENCH_SHAPES=$OUT/shapes_cfg3.md python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline --clips 0 > $OUT/bench_cfg3.json 2>> $OUT/bench.err
EMO_BENCH_SHAPES=$OUT/shapes_cfg5.md python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --clips 0 > $OUT/bench_cfg5.json 2>> $OUT/bench.err
python bench.py --mode strong --steps 10 --warmup 3 --no-cpu-baseline --no-profile --clips 0 > $OUT/bench_strong_cfg4_1gpu.json 2>> $OUT/bench.err
python bench.py --controlnet --steps 10 --warmup 3 --no-cpu-baseline --no-profile --clips 0 > $OUT/bench_controlnet.json 2>> $OUT/bench.err
python bench.py --stage vae --steps 5 --warmup 2 > $OUT/bench_vae.json 2>> $OUT/bench.err
python bench.py --mode strong --emulate-rank 4/8 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --clips 0 > $OUT/emu_strong_cond.json 2>> $OUT/bench.err
python bench.py --emulate-rank 0/8 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --clips 0 > $OUT/emu_weak8.json 2>> $OUT/bench.err
python bench.py --gpus 2 --share-gpu --steps 10 --warmup 3 --no-cpu-baseline --no-profile --clips 0 > $OUT/share_gpu_2ranks.json 2>> $OUT/bench.err
ls -la $OUT
