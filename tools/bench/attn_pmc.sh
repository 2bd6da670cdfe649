#!/bin/bash
# usage on the GPU box: tools/bench/attn_pmc.sh OUTFILE [lib.so ...]   - SQ / TCP counters of the attention kernel per library
export TMPDIR=/tmp
OUT=$1; shift
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P5="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
P6="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES"
: > $OUT
for lib in "${@:-product}"; do
  if [ "$lib" != "product" ]; then export EMO_HIP_LIB=$PWD/$lib; else unset EMO_HIP_LIB; fi
  echo "=== $lib" >> $OUT
  for i in 1 2 3 4 5 6; do
    eval PP=\$P$i; rm -rf /tmp/apmc$i
    rocprofv3 --kernel-trace --pmc $PP --output-format csv -d /tmp/apmc$i -o p -- python tools/bench/attn_pmc.py > /tmp/apmc$i.log 2>&1
  done
  python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "a")
for i in range(1, 7):
    d = f"/tmp/apmc{i}"
    fs = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not fs:
        out.write(f"{d}: no counter csv: " + open(f"/tmp/apmc{i}.log").read()[-300:].replace("\n", " | ") + "\n"); continue
    acc = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(fs[0])):
        if "attention_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        acc["_vgpr"] = float(r.get("VGPR_Count", 0) or 0); acc["_lds"] = float(r.get("LDS_Block_Size", 0) or 0)
    nd = max(1, len(disp))
    for k, v in sorted(acc.items()):
        out.write(f"    {k:32s} {v / (1 if k.startswith('_') else nd):18.0f}\n")
out.close()
PY
done
cat $OUT
