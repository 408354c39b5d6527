"""Gaps between consecutive kernels of each HIP queue in a rocprofv3 --kernel-trace CSV.

    python tools/gap_report.py <kernel_trace.csv> [t_lo_frac t_hi_frac]

Prints, per queue, busy time, idle time, and the idle time grouped by the kernel that FOLLOWS the gap -- which shows
whether several passes in flight lose their time inside kernels (contention) or between them (dispatch)."""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("dec_cross_attn", "dec_self_attn", "skinny_kernel<true", "skinny_kernel<false, 10", "skinny_kernel<false, 5",
                "skinny2", "skinny_mt", "sample_kernel", "gemm_8phase", "attn_full", "embed", "advance", "resln", "layernorm",
                "dtw_", "logmel", "set_ints"):
        if key in name:
            return key
    return name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    by_q = defaultdict(list)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= a and e <= b:
            by_q[r["Queue_Id"]].append((s, e, short(r["Kernel_Name"])))
    for q, ks in sorted(by_q.items()):
        ks.sort()
        if len(ks) < 1000:
            continue
        busy = sum(e - s for s, e, _ in ks)
        gaps = defaultdict(lambda: [0, 0])
        idle = 0
        for (s0, e0, _), (s1, e1, n1) in zip(ks, ks[1:]):
            g = max(0, s1 - e0)
            idle += g
            gaps[n1][0] += g
            gaps[n1][1] += 1
        span = ks[-1][1] - ks[0][0]
        print(f"queue {q}: {len(ks)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms")
        for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:10]:
            print(f"    before {n:28s} {g / 1e6:8.2f} ms in {c:6d} gaps ({g / max(c, 1) / 1e3:6.2f} us each)")


if __name__ == "__main__":
    main()
