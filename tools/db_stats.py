"""per-kernel statistics from a rocprofv3 results .db (rocpd sqlite): name, calls, total / average / min duration"""
import sqlite3, sys, re
for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(path, "total kernel time %.1f ms" % (tot / 1e6))
    for name, n, s, a, mn in rows[:18]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
        print(f"  {short:70s} {n:7d} {s / 1e6:9.2f} ms {100 * s / tot:5.1f}%  avg {a / 1e3:8.2f} us  min {mn / 1e3:8.2f} us")
