#!/usr/bin/env python3
"""Per-queue (stream) timeline of one training step from a rocprofv3 kernel trace: busy time, gaps,
per-kernel totals on the critical (main) queue."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
ad = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
a, b = ad[-3], ad[-2]
seg = rows[a + 1:b + 1]
print("step wall ms %.3f  kernels %d" % ((seg[-1][2] - seg[0][1]) / 1e6, len(seg)))
byq = defaultdict(list)
for r in seg:
    byq[r[3]].append(r)
main = max(byq, key=lambda q: len(byq[q]))
for q, rs in byq.items():
    busy = sum(r[2] - r[1] for r in rs)
    gaps = [rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print("queue %s%s n %d busy %.3f ms gaps %.3f ms" % (q, " (main)" if q == main else "", len(rs), busy / 1e6, sum(pos) / 1e6))
agg = defaultdict(lambda: [0, 0])
for r in byq[main]:
    k = r[0].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
    agg[k][0] += 1
    agg[k][1] += r[2] - r[1]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-72s %5d %8.3f ms %7.1f us" % (k, n, t / 1e6, t / n / 1e3))
