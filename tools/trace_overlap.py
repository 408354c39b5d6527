"""What several passes in flight do to each other, from a rocprofv3 --kernel-trace CSV of the product configuration.

    python tools/trace_overlap.py <dir holding *_kernel_trace.csv> [t_lo_frac t_hi_frac | --inflight N]

--inflight N (default 4): only the part of the trace in which N queues are busy at once (10 ms buckets in which N
queues each run >= 100 kernels): the timed region of bench.py with N passes in flight, without warm-up, the
single-stream probes and the idle stretches around them.

Prints:
  * how busy the GPU is: union of all kernel intervals / span, and the mean number of kernels running at once;
  * per kernel family: launches, average duration, share of the summed kernel time;
  * time during which at least one encoder kernel (GEMM / attention / LayerNorm / log-mel) runs, and how much of the
    decode kernels' time falls inside it -- whether encode and decode of different passes overlap;
  * per queue: busy / idle."""
import csv
import glob
import os
import sys
from collections import defaultdict

FAMILIES = [("dec_cq_xattn", "dec_cq_xattn"), ("dec_cross_attn", "dec_cross_attn"), ("dec_self_attn", "dec_self_attn"),
            ("dec_hfused", "dec_hfused"), ("skinny2", "logits"), ("ln_rows16", "logits"), ("skinny_vw2", "gemv_fc2"), ("skinny_wide_kernel<16", "gemv_fc2"), ("skinny_wide_kernel", "gemv_wide"), ("ln_rows32_blk", "gemv_wide_ln"), ("skinny_kernel<true", "gemv_ln"), ("skinny_kernel<false, 10", "gemv_fc2"),
            ("skinny_kernel<false", "gemv_proj"), ("skinny", "gemv_other"), ("sample_kernel", "sampler"), ("gemm_8phase", "enc_gemm256"), ("gemm_pipe", "enc_gemm256"),
            ("gemm_glds", "enc_gemm128"), ("attn_full", "enc_attention"), ("layernorm", "enc_layernorm"), ("logmel", "logmel"), ("dtw", "dtw")]
ENCODER = {"enc_gemm256", "enc_gemm128", "enc_attention", "enc_layernorm", "logmel"}


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def union(iv):
    iv = sorted(iv)
    out, tot = [], 0
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out, sum(e - s for s, e in out)


def overlap_with(merged, s, e):
    """length of [s, e) covered by the merged (sorted, disjoint) interval list"""
    import bisect
    i = bisect.bisect_right(merged, [s, 1 << 62]) - 1
    tot = 0
    i = max(i, 0)
    while i < len(merged) and merged[i][0] < e:
        tot += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
        i += 1
    return tot


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(d) else [d]
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), family(r["Kernel_Name"]), r["Queue_Id"]) for r in rows]
    t0 = min(k[0] for k in ks)
    t1 = max(k[1] for k in ks)
    if len(sys.argv) > 3 and sys.argv[2] != "--inflight":
        lo, hi = float(sys.argv[2]), float(sys.argv[3])
        a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
        ks = [k for k in ks if k[0] >= a and k[1] <= b]
        span = max(k[1] for k in ks) - min(k[0] for k in ks)
    else:
        n_q = int(sys.argv[3]) if len(sys.argv) > 3 else 4
        BUCKET = 10_000_000
        per = defaultdict(lambda: defaultdict(int))
        for s_, e_, f_, q_ in ks:
            per[(s_ - t0) // BUCKET][q_] += 1
        good = {bk for bk, qs in per.items() if sum(1 for v in qs.values() if v >= 100) >= n_q}
        ks = [k for k in ks if (k[0] - t0) // BUCKET in good and (k[1] - t0) // BUCKET in good]
        span = len(good) * BUCKET
        print(f"{len(good)} buckets of 10 ms with {n_q} queues busy")
    merged, busy = union([(s, e) for s, e, _, _ in ks])
    total = sum(e - s for s, e, _, _ in ks)
    print(f"window: {span / 1e6:.1f} ms, {len(ks)} kernels on {len(set(k[3] for k in ks))} queues")
    print(f"GPU busy (union of kernel intervals): {busy / span:.3f} of the window; kernels running at once (mean over busy time): {total / busy:.2f}")
    fam = defaultdict(lambda: [0, 0])
    for s, e, f, _ in ks:
        fam[f][0] += 1
        fam[f][1] += e - s
    print(f"{'family':16s} {'launches':>9s} {'avg us':>9s} {'share':>7s}")
    for f, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{f:16s} {n:9d} {t / n / 1e3:9.2f} {t / total:7.3f}")
    enc_m, enc_busy = union([(s, e) for s, e, f, _ in ks if f in ENCODER])
    dec = [(s, e) for s, e, f, _ in ks if f not in ENCODER]
    dec_m, dec_busy = union(dec)
    both = sum(overlap_with(enc_m, s, e) for s, e in dec_m)
    print(f"encoder kernels cover {enc_busy / span:.3f} of the window, decode kernels {dec_busy / span:.3f}, both at once {both / span:.3f}")
    by_q = defaultdict(list)
    for s, e, f, q in ks:
        by_q[q].append((s, e, f))
    for q, iv in sorted(by_q.items()):
        if len(iv) < 500:
            continue
        _, qb = union([(s, e) for s, e, _ in iv])
        iv.sort()
        gaps = defaultdict(lambda: [0, 0])
        for (s0, e0, _), (s1, e1, f1) in zip(iv, iv[1:]):
            g = s1 - e0
            if 0 <= g < 200_000:                 # gaps between dependent kernels, not the pauses between passes
                gaps[f1][0] += g
                gaps[f1][1] += 1
        tot_g = sum(g for g, _ in gaps.values())
        tot_n = sum(n for _, n in gaps.values())
        print(f"queue {q}: {len(iv)} kernels, busy {qb / span:.3f} of the window, mean gap in front of a kernel {tot_g / max(tot_n, 1) / 1e3:.2f} us")
        for f, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:6]:
            print(f"      before {f:14s} {g / max(n, 1) / 1e3:6.2f} us x {n}")


if __name__ == "__main__":
    main()
