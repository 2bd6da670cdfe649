#!/bin/bash
# Which hipBLASLt kernels serve the path's dense shapes (full Tensile names: macro tile, wave grid, prefetch depth, LDS use)?
# usage on the GPU box: tools/bench/library_kernels.sh OUTFILE
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lib -o lib -- python tools/bench/library_yardstick.py --dense-only > /tmp/lib.log 2>&1
python - "$1" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/prof_lib/**/lib_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if n.startswith("Cijk") or "gemm_kernel" in n:
        d[(n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[1], "w") as o:
    for (n, g, w, lds, vg, ag), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        o.write(f"{len(v):5d} x {sum(v)/len(v)/1e3:8.1f} us  grid {g} wg {w} lds {lds} vgpr {vg} agpr {ag}  {n}\n")
PY
