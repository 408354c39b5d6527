"""One row per kernel AND launch width from a rocprofv3 --kernel-trace CSV of bench.py.

    python tools/trace_by_width.py <dir holding *_kernel_trace.csv> <out.csv> [--inflight N]

The per-kernel --stats summary folds every launch of a kernel into one average: the 16-row launches of the command's
batch-16 phase, the wide launches of the `value` run, warm-up, the ramp of a job (fewer passes in flight: a launch has more
of the HBM) and its steady state.  Here dispatches are grouped by (kernel, workgroups in the grid); for the decode kernels
whose grid follows the launch's rows the width is spelled out (Rows).  Two averages per row: over the whole trace, and
inside the part of the trace where >= N queues are busy (N passes in flight, 10 ms buckets with >= 100 kernels on each of
N queues) -- the steady state the live launch timer of bench.py measures.
Columns: Name, Workgroups, Rows, Calls, AverageNs, InFlightCalls, InFlightAverageNs."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")
    name = re.sub(r"^void\s+", "", name)
    if name.startswith("_Z"):                         # mangled: _ZN12_GLOBAL__N_120dec_self_attn_kernelE...
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)", name)
        return m.group(1) if m else name[:60]
    depth, out = 0, []
    for ch in name:                                   # drop the argument list: the last top-level "(...)"
        if ch == "(" and depth == 0 and out and "".join(out).strip():
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def rows_of(name, wgs, gz):
    """launch rows of the decode kernels whose grid is a function of the rows (large-v3: d = 1280, 20 heads)"""
    if "dec_cq_xattn" in name:                        # (rows / 16) x 160 GEMV blocks + rows x 20 attention blocks
        return wgs / 30.0 if wgs % 30 == 0 else -1
    if "dec_self_attn" in name:                       # one block per (row, head)
        return wgs / 20.0 if wgs % 20 == 0 else -1
    return -1


def main():
    d, out = sys.argv[1], sys.argv[2]
    n_q = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--inflight" else 3
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(d) else [d]
    ks = []
    for f in files:
        for r in csv.DictReader(open(f)):
            wg = 1
            for ax in "XYZ":
                g, w = int(r.get(f"Grid_Size_{ax}", 1) or 1), int(r.get(f"Workgroup_Size_{ax}", 1) or 1)
                wg *= max(1, g // max(1, w))
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], wg,
                       int(r.get("Grid_Size_Y", 1) or 1)))
    t0 = min(k[0] for k in ks)
    BUCKET = 10_000_000
    per = defaultdict(lambda: defaultdict(int))
    for s_, e_, _n, q_, _w, _y in ks:
        per[(s_ - t0) // BUCKET][q_] += 1
    good = {bk for bk, qs in per.items() if sum(1 for v in qs.values() if v >= 100) >= n_q}
    agg = defaultdict(lambda: [0, 0, 0, 0])
    for s_, e_, n_, _q, w_, y_ in ks:
        a = agg[(n_, w_)]
        a[0] += 1
        a[1] += e_ - s_
        if (s_ - t0) // BUCKET in good and (e_ - t0) // BUCKET in good:
            a[2] += 1
            a[3] += e_ - s_
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Name", "Workgroups", "Rows", "Calls", "AverageNs", "InFlightCalls", "InFlightAverageNs"])
        for (n_, w_), (c, t, ci, ti) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            r = rows_of(n_, w_, 1)
            wr.writerow([n_, w_, int(r) if r == int(r) else -1, c, round(t / c, 1), ci, round(ti / ci, 1) if ci else ""])
    print(f"{len(agg)} (kernel, width) rows from {len(ks)} dispatches; in-flight window: {len(good)} buckets of 10 ms with >= {n_q} queues busy -> {out}")


if __name__ == "__main__":
    main()
