#!/usr/bin/env python3
"""Per (kernel, grid size) means of the PMC counters in a rocprofv3 run_results.db: usage pmc_by_grid.py DB [name-filter]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
ki, ci, vi, gi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("grid_size")
agg = collections.defaultdict(lambda: [0.0, 0])
for r in cur.execute("select * from counters_collection"):
    if flt and flt not in r[ki]:
        continue
    name = r[ki].split("(anonymous namespace)::")[-1].split("(")[0]
    a = agg[(name, int(r[gi]), r[ci])]
    a[0] += float(r[vi]); a[1] += 1
for (name, grid, c), (v, n) in sorted(agg.items()):
    print("%-70s grid %8d %-28s n %3d mean %14.1f" % (name[:70], grid, c, n, v / n))
