#!/usr/bin/env python3
"""Idle-gap analysis of a rocprofv3 --kernel-trace database: for the last loop iterations (delimited by the
cfg_step kernel) report wall time, the union of kernel-busy intervals, the number of dispatches, the largest gaps between
consecutive kernels and the time spent in kernels that are NOT this library's (torch index / copy plumbing, rocBLAS)."""
import re
import sqlite3
import sys

FOREIGN = re.compile(r"at::|c10::|rocblas|Cijk_|hipblas|nccl|rccl|thrust|cub::")   # everything else is a libemo_hip.so kernel
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "cfg_step_kernel" in r[0]]
if len(marks) < 3:
    raise SystemExit("need >= 3 steps in the trace")
for a, b in zip(marks[-3:-1], marks[-2:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for n, s, e in sorted((r for r in seg), key=lambda r: r[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(r[2] - r[1] for r in seg)
    foreign = [(r[0], r[2] - r[1]) for r in seg if FOREIGN.search(r[0])]
    print(f"step: {len(seg)} dispatches, wall {(t1 - t0) / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, "
          f"sum of kernel durations {tot / 1e6:.3f} ms; foreign (torch / rocBLAS) kernels: {len(foreign)} dispatches, "
          f"{sum(d for _, d in foreign) / 1e6:.3f} ms")
    gaps.sort(reverse=True)
    print("  largest gaps (us, before kernel): " + ", ".join(f"{g / 1e3:.1f} {re.sub(r'[(<].*$', '', n)[:40]}" for g, n in gaps[:6]))
    agg = {}
    for n, d in foreign:
        k = re.sub(r"\(.*$", "", n)[:70]
        agg[k] = agg.get(k, 0) + d
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1])[:6]:
        print(f"  foreign: {d / 1e3:8.1f} us  {k}")
