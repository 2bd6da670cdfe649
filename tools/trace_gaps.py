#!/usr/bin/env python3
"""Idle-gap analysis of a rocprofv3 --kernel-trace database: for the last `steps` loop iterations (delimited by the
cfg_step kernel) report wall time, the union of kernel-busy intervals and the number of dispatches."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "cfg_step_kernel" in r[0]]
if len(marks) < 3:
    raise SystemExit("need >= 3 steps in the trace")
for a, b in zip(marks[-3:-1], marks[-2:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    busy, cur_s, cur_e = 0, None, None
    for _, s, e in sorted((r for r in seg), key=lambda r: r[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(r[2] - r[1] for r in seg)
    print(f"step: {len(seg)} dispatches, wall {(t1 - t0) / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, "
          f"sum of kernel durations {tot / 1e6:.2f} ms")
