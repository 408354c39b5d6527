#!/usr/bin/env python3
"""The request-size split behind FETCH_SIZE (MI355X_MICROARCH.md, HBM section: FETCH_SIZE = TCC_EA0_RDREQ x 64 B, with
128-B requests tallied at 64 B): per kernel, median TCC_EA0_RDREQ_sum (all read requests), TCC_EA0_RDREQ_32B_sum (the
32-byte ones) and FETCH_SIZE from three separate rocprofv3 --pmc passes.  Usage: pmc_requests.py RDREQ_DIR RDREQ32_DIR FETCH_DIR"""
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
    return {k: statistics.median(v) for k, v in acc.items()}


rd, rd32, fs = medians(sys.argv[1], "TCC_EA0_RDREQ_sum"), medians(sys.argv[2], "TCC_EA0_RDREQ_32B_sum"), medians(sys.argv[3], "FETCH_SIZE")
out = {}
for k in sorted(set(rd) & set(fs)):
    if "at::native" in k:
        continue
    n, n32, f = rd[k], rd32.get(k, 0.0), fs[k] * 1024
    out[k] = {"TCC_EA0_RDREQ": n, "TCC_EA0_RDREQ_32B": n32, "FETCH_SIZE_bytes": f,
              "FETCH_SIZE_per_request_B": round(f / n, 2) if n else None,
              "bytes_if_non32B_requests_are_64B": n32 * 32 + (n - n32) * 64,
              "bytes_if_non32B_requests_are_128B": n32 * 32 + (n - n32) * 128}
print(json.dumps(out, indent=1))
