#!/usr/bin/env python3
"""Turn two rocprofv3 PMC databases (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, same command) into
profiles/rNN_pmc.json: HBM-side bytes per launch for each kernel family bench.py reports.

    python tools/pmc_to_json.py fetch_results.db write_results.db > profiles/r01e_pmc.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes, so it is doubled
(MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is uncalibrated (reported as is)."""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emote_hack_amd.build import csrc_digest  # noqa: E402  (the build the counters belong to)

FAMILY = [("layernorm_stats_kernel", "layernorm_stats"), ("softmax_rows_kernel", "softmax_rows"), ("conv3x3_halo_kernel", "gemm_conv3x3"), (r"gemm_kernel<[^>]*?(unsigned short|float), true", "gemm_conv3x3"),
          ("gemm_kernel", "gemm_dense"), ("gemm_splitk_epilogue", "gemm_splitk_epilogue"), ("temporal_attention_mfma_kernel", "temporal_attention"), ("temporal_attention_kernel", "temporal_attention"),
          ("attention_kernel", "attention"), ("layernorm_kernel", "layernorm"), ("gn_stats_kernel", "groupnorm"), ("gn_apply_kernel", "groupnorm"), ("gn_one_kernel", "groupnorm"), ("gn_fold_linear_kernel", "groupnorm")]


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


def mfma(path):
    """MFMA utilisation (half of BASELINE.json's metric) from one pass with --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
    GRBM_GUI_ACTIVE.  SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles with the matrix pipe busy (32 per v_mfma_f32_32x32x16_bf16,
    MI355X_MICROARCH.md constants table; checked against the conv FLOPs of the run).  rocprofv3 on gfx950 reports every counter
    once PER XCD and dispatch (8 rows per launch: 8 x 1050 dense launches = 8400 rows), each GRBM_GUI_ACTIVE row holding the
    dispatch's full duration in shader cycles, so
        util = sum(MFMA_BUSY rows) / (sum(GUI_ACTIVE rows) / XCDS * 256 CUs * 4 SIMDs) = sum(busy) / (sum(gui) * 128)."""
    XCDS, SIMDS = 8, 1024
    db = sqlite3.connect(path)
    per = {}
    for name, cn, v in db.execute("select name, counter_name, counter_value from pmc_events"):
        f = family(name) or "other"
        d = per.setdefault(f, {})
        d[cn] = d.get(cn, 0.0) + float(v)
        d["_n_" + cn] = d.get("_n_" + cn, 0) + 1
    out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 1 --warmup 1 "
                   "--no-graphs --no-profile --no-cpu-baseline` (eager: the ReferenceNet group pass weighs more than in a 50-step run); "
                   "counters arrive once per XCD and dispatch; util = sum(MFMA_BUSY) / (sum(GRBM_GUI_ACTIVE) / 8 * 1024 SIMDs)",
           "per_kernel_family": {}}
    tot_b = tot_a = 0.0
    for f, d in sorted(per.items()):
        b, a = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
        out["per_kernel_family"][f] = {"launches": d.get("_n_GRBM_GUI_ACTIVE", 0) // XCDS, "mfma_busy_cycles": b,
                                       "dispatch_cycles": a / XCDS, "sq_busy_cu_cycles": d.get("SQ_BUSY_CU_CYCLES", 0.0),
                                       "mfma_busy_frac": b / (a / XCDS * SIMDS) if a else None}
        if f != "other":   # torch's own kernels (weight synthesis, packing) are not the path
            tot_b += b
            tot_a += a
    out["mfma_busy_frac"] = tot_b / (tot_a / XCDS * SIMDS) if tot_a else None
    out["csrc_sha256"] = csrc_digest()
    out["mfma_busy_frac_note"] = "over every dispatch of the path's kernel families (GEMM, conv, attention AND the HBM-bound kernels), weighted by time in flight"
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1] == "--mfma":
        return mfma(sys.argv[2])
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --no-graphs "
                   "--no-profile --no-cpu-baseline` (2 loop iterations), gfx950 correction FETCH x2 (MI355X_MICROARCH.md HBM section); "
                   "WRITE_SIZE uncalibrated; GroupNorm = stats + apply launches",
           "csrc_sha256": csrc_digest(), "per_kernel_family": {}}
    for fam in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(fam, [0, 0.0])
        nw, w = write.get(fam, [0, 0.0])
        n = max(nf, nw, 1)
        out["per_kernel_family"][fam] = {"launches": n, "fetch_kb_raw_per_launch": f / max(nf, 1), "write_kb_raw_per_launch": w / max(nw, 1),
                                         "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
