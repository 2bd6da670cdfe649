#!/bin/bash
# usage on the GPU box: tools/bench/path_pmc.sh OUTFILE [subject.py]  - SQ counters per (kernel, grid) of the subject's launches
export TMPDIR=/tmp
OUT=$1; SUBJ=${2:-tools/bench/path_pmc.py}
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"
for i in 1 2; do
  eval PP=\$P$i; rm -rf /tmp/ppmc$i
  rocprofv3 --kernel-trace --pmc $PP --output-format csv -d /tmp/ppmc$i -o p -- python $SUBJ > /tmp/ppmc$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1], "w")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); order = []
for i in (1, 2):
    fs = glob.glob(f"/tmp/ppmc{i}/**/p_counter_collection.csv", recursive=True)
    if not fs:
        out.write(f"pass {i}: no counter csv: " + open(f"/tmp/ppmc{i}.log").read()[-300:].replace("\n", " | ") + "\n"); continue
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if not ("gemm_kernel" in n or "conv3x3_halo" in n or "attention_kernel" in n or "temporal" in n):
            continue
        key = (n.split("(")[0][:70], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        if key not in order: order.append(key)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(key, i)].add(r["Dispatch_Id"])
for key in order:
    nd = max(1, len(disp[(key, 1)]) or len(disp[(key, 2)]))
    c = {k: v / nd for k, v in acc[key].items()}
    simd = c.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024 or 1          # SIMD-cycles of the dispatch (8 XCDs report the same interval)
    q = lambda k: 4 * c.get(k, 0)                                   # quad-cycle counters -> cycles
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    out.write(f"{key[0]} grid {key[1]} lds {key[2]} x{nd}\n")
    out.write(f"    SIMD-cycles {simd / 1e6:9.1f} M | active: MFMA {mfma / simd:5.1%}  VALU(other) {(q('SQ_ACTIVE_INST_VALU') - mfma) / simd:5.1%}  SALU {q('SQ_ACTIVE_INST_SCA') / simd:5.1%}"
              f"  LDS {q('SQ_ACTIVE_INST_LDS') / simd:5.1%}  MISC {q('SQ_ACTIVE_INST_MISC') / simd:5.1%}  VMEM {q('SQ_ACTIVE_INST_VMEM') / simd:5.1%}  | sum {q('SQ_ACTIVE_INST_ANY') / simd:5.1%}\n")
    out.write(f"    per MFMA: VALU {(c.get('SQ_INSTS_VALU', 0) - c.get('SQ_INSTS_MFMA', 0)) / max(1, c.get('SQ_INSTS_MFMA', 1)):5.2f}  SALU {c.get('SQ_INSTS_SALU', 0) / max(1, c.get('SQ_INSTS_MFMA', 1)):5.2f}"
              f"  LDS {c.get('SQ_INSTS_LDS', 0) / max(1, c.get('SQ_INSTS_MFMA', 1)):5.2f} | wave: active {c.get('SQ_ACTIVE_INST_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):5.1%}"
              f"  issue-stall {c.get('SQ_WAIT_INST_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):5.1%}  parked {c.get('SQ_WAIT_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):5.1%}\n")
out.close()
PY
cat $OUT
