#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: median counter value per kernel name."""
import csv
import glob
import statistics
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):4d} median={statistics.median(v):.4g}")
