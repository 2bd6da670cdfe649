#!/usr/bin/env python3
"""Turn two rocprofv3 PMC databases (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, same command) into
profiles/rNN_pmc.json: HBM-side bytes per launch for each kernel family bench.py reports.

    python tools/pmc_to_json.py fetch_results.db write_results.db > profiles/r01e_pmc.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes, so it is doubled
(MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is uncalibrated (reported as is)."""
import json
import re
import sqlite3
import sys

FAMILY = [("conv3x3_halo_kernel", "gemm_conv3x3"), (r"gemm_kernel<[^>]*?(unsigned short|float), true", "gemm_conv3x3"),
          ("gemm_kernel", "gemm_dense"), ("gemm_splitk_epilogue", "gemm_splitk_epilogue"), ("temporal_attention_kernel", "temporal_attention"),
          ("attention_kernel", "attention"), ("layernorm_kernel", "layernorm"), ("gn_stats_kernel", "groupnorm"), ("gn_apply_kernel", "groupnorm")]


def family(name):
    for pat, fam in FAMILY:
        if re.search(pat, name):
            return fam
    return None


def load(path, counter):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    if "pmc_events" in tabs:
        rows = db.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)).fetchall()
    else:
        raise SystemExit(f"{path}: no pmc_events view (tables: {tabs[:20]})")
    agg = {}
    for name, v in rows:
        f = family(name)
        if f is None:
            continue
        d = agg.setdefault(f, [0, 0.0])
        d[0] += 1
        d[1] += float(v)
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --no-graphs "
                   "--no-profile --no-cpu-baseline` (2 loop iterations), gfx950 correction FETCH x2 (MI355X_MICROARCH.md HBM section); "
                   "WRITE_SIZE uncalibrated; GroupNorm = stats + apply launches",
           "per_kernel_family": {}}
    for fam in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(fam, [0, 0.0])
        nw, w = write.get(fam, [0, 0.0])
        n = max(nf, nw, 1)
        out["per_kernel_family"][fam] = {"launches": n, "fetch_kb_raw_per_launch": f / max(nf, 1), "write_kb_raw_per_launch": w / max(nw, 1),
                                         "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
