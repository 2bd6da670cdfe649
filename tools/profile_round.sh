#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh r02d
# rocprofv3 passes over the SAME command (bench.py, cfg2, bf16): kernel trace -> per-kernel stats; PMC FETCH_SIZE, PMC WRITE_SIZE
# and the MFMA-busy SQ counters each in their OWN pass (MI355X_MICROARCH.md: TCC slots; gpurun refuses --pmc with sys traces).
# Summaries land in gpurun_out/<tag>/ - copy the ones to keep into profiles/.
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 2 --no-profile --no-cpu-baseline --clips 0"
PMC_CMD="python bench.py --steps 1 --warmup 1 --no-graphs --no-profile --no-cpu-baseline --clips 0"
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $CMD > $OUT/kt.log 2>&1
python tools/rocpd_summary.py $(ls /tmp/prof_kt/*/kt_results.db /tmp/prof_kt/kt_results.db 2>/dev/null | head -1) > $OUT/kernel_stats.md 2>> $OUT/kt.log
python tools/trace_gaps.py $(ls /tmp/prof_kt/*/kt_results.db /tmp/prof_kt/kt_results.db 2>/dev/null | head -1) > $OUT/trace_gaps.txt 2>> $OUT/kt.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $PMC_CMD > $OUT/pmc_write.log 2>&1
python tools/pmc_to_json.py $(ls /tmp/prof_f/*/f_results.db /tmp/prof_f/f_results.db 2>/dev/null | head -1) \
                            $(ls /tmp/prof_w/*/w_results.db /tmp/prof_w/w_results.db 2>/dev/null | head -1) > $OUT/pmc.json 2>> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_m -o m -- $PMC_CMD > $OUT/pmc_mfma.log 2>&1
python tools/pmc_to_json.py --mfma $(ls /tmp/prof_m/*/m_results.db /tmp/prof_m/m_results.db 2>/dev/null | head -1) > $OUT/mfma.json 2>> $OUT/pmc_mfma.log
tail -3 $OUT/kt.log; head -c 600 $OUT/mfma.json
