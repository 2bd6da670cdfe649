#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) --kernel-trace database as a per-kernel stats table:
    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {v[0]} | {v[1] / 1e3:.3f} | {v[1] / v[0]:.2f} | {v[2]:.2f} | {v[3]:.2f} | {100 * v[1] / tot:.1f} |")
    print(f"\ntotal kernel time {tot / 1e3:.3f} ms over {sum(v[0] for v in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])


def pmc(path, counter):
    """per-kernel average of a PMC counter from a `rocprofv3 --pmc <counter> --kernel-trace` database"""
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)).fetchall()
    agg = {}
    for name, v in rows:
        d = agg.setdefault(short(name), [0, 0.0])
        d[0] += 1; d[1] += v
    return agg
