#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 run_results.db (one --pmc pass)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("# tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()][:12])
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    sys.exit("no counters_collection view")
cols = [d[1] for d in cur.execute(f"pragma table_info({view})")]
print("# columns:", cols)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
for r in cur.execute(f"select * from {view}"):
    a = agg[r[ki]][r[ci]]
    a[0] += float(r[vi]); a[1] += 1
for k, d in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    for c, (v, n) in d.items():
        print(f"{k[:90]:90s} {c:28s} dispatches {n:6d} sum {v:16.0f} per-dispatch {v / max(n, 1):14.1f}")
