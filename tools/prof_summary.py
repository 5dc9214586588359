#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run_results.db.
usage: prof_summary.py <db> <steps> [last N]   -- with "last N": only the kernels of the last N optimizer steps of the
trace (steps are delimited by sumsq_kernel, the one launch per update on the data-path stream -- the Adam kernel runs in
chunks beside the next forward since round 6): bench.py replays its timed steps at the end with per-dispatch events, so
"last <steps>" is exactly the set of launches behind the bench line's roofline figures."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
last = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "last" else 0
rows = list(db.execute("select name, start, end from kernels order by start"))
if last:
    ad = [i for i, r in enumerate(rows) if "sumsq_kernel" in r[0]]
    lo = ad[-last - 1] + 1 if len(ad) > last else 0
    rows = rows[lo:ad[-1] + 1]
    steps = float(last)
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[n]
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"# {'the last %d steps of the trace' % last if last else 'all kernels'}: {tot / 1e3 / steps:.3f} ms of kernel time per step over {steps:g} steps (durations in us)")
print(f"{'kernel':100s} {'calls':>7s} {'ms/step':>9s} {'avg_us':>9s} {'share':>6s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:100]:100s} {c:7d} {t / 1e3 / steps:9.3f} {t / c:9.2f} {100 * t / tot:6.2f}")
