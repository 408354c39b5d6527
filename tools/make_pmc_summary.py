#!/usr/bin/env python3
"""Builds profiles/rNN_pmc_summary.json from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md's HBM section prescribes).  Usage:
    make_pmc_summary.py OUT.json LABEL=ROWS:FETCH_DIR:WRITE_DIR [...]
Per kernel: median FETCH_SIZE / WRITE_SIZE (KB = 1024 B) and HBM bytes per launch with the gfx950 correction
(FETCH_SIZE x2 for wide coalesced 16-byte loads).  The uncorrected sum is kept beside it: for the LDS-DMA
(global_load_lds) staged GEMM the raw counter already matches the algorithmic bytes and x2 would over-count."""
import csv
import glob
import json
import statistics
import sys
from collections import defaultdict


def medians(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (statistics.median(v), len(v)) for k, v in acc.items()}


def ours(name):
    return ("anonymous namespace)::" in name or "_GLOBAL__N_" in name) and "at::native" not in name


out = {"how": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python tools/probe_kernels.py ...; "
              "gfx950 correction: FETCH_SIZE x2 for wide coalesced 16-B loads (MI355X_MICROARCH.md HBM section); KB = 1024 B",
       "kernels": {}}
for spec in sys.argv[2:]:
    label, rest = spec.split("=", 1)
    rows, fdir, wdir = rest.split(":")
    F, W = medians(fdir, "FETCH_SIZE"), medians(wdir, "WRITE_SIZE")
    for k in sorted(set(F) | set(W)):
        if not ours(k):
            continue
        f, n = F.get(k, (0.0, 0))
        w, _ = W.get(k, (0.0, 0))
        out["kernels"][f"{label}: {k}"] = {"rows": int(rows), "launches": n, "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w,
                                           "hbm_bytes_per_launch_corrected": int(round((2 * f + w) * 1024)),
                                           "hbm_bytes_per_launch_uncorrected": int(round((f + w) * 1024))}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in out["kernels"].items()}, indent=1))
