#!/bin/bash
# usage on the GPU box: tools/bench/gemm_pmc2.sh OUTFILE
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"
P4="VmemLatency LdsLatency"
P5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"
P6="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
for i in 4 5 6; do eval PP=\$P$i; rocprofv3 --kernel-trace --pmc $PP --output-format csv -d /tmp/pmc$i -o p -- python tools/bench/gemm_pmc2.py > /tmp/pmc$i.log 2>&1; done
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d /tmp/pmc1 -o p -- python tools/bench/gemm_pmc2.py > /tmp/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d /tmp/pmc2 -o p -- python tools/bench/gemm_pmc2.py > /tmp/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc $P3 --output-format csv -d /tmp/pmc3 -o p -- python tools/bench/gemm_pmc2.py > /tmp/pmc3.log 2>&1
python - "$1" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
for d in ("/tmp/pmc1", "/tmp/pmc2", "/tmp/pmc3", "/tmp/pmc4", "/tmp/pmc5", "/tmp/pmc6"):
    fs = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not fs:
        out.write(f"{d}: no counter csv\n"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); disp = {}
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if not ("Cijk" in n or "gemm_kernel" in n):
            continue
        key = (n[:120], r.get("Grid_Size", "") + "/" + r.get("LDS_Block_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp.setdefault(key, set()).add(r["Dispatch_Id"])
    for key, c in acc.items():
        nd = len(disp[key])
        out.write(f"{key[0]} grid {key[1]} dispatches {nd}\n")
        for k, v in sorted(c.items()):
            out.write(f"    {k:28s} {v / nd:16.0f} per dispatch\n")
out.close()
PY
tail -2 /tmp/pmc1.log
