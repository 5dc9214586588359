#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace --stats run_results.db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"# all kernels: {tot / 1e3 / steps:.3f} ms per step over {steps:g} steps (durations in us)")
print(f"{'kernel':100s} {'calls':>7s} {'ms/step':>9s} {'avg_us':>9s} {'share':>6s}")
for n, c, t, a, p in rows:
    print(f"{n[:100]:100s} {c:7d} {t / 1e3 / steps:9.3f} {a:9.2f} {p:6.2f}")
