#!/usr/bin/env python3
"""Per-queue (stream) timeline of one training step from a rocprofv3 kernel trace: busy time, gaps,
per-kernel totals on the critical (main) queue."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
# (steps are delimited by the gradient-norm pass, one launch per update on the data-path stream: since round 6 the Adam kernel
#  runs in chunks on the second stream beside the NEXT step's forward -- those chunks belong to the window they run in)
ad = [i for i, r in enumerate(rows) if 'sumsq_kernel' in r[0]]
a, b = ad[-3], ad[-2]
seg = rows[a + 1:b + 1]
print("step wall ms %.3f  kernels %d" % ((seg[-1][2] - seg[0][1]) / 1e6, len(seg)))
byq = defaultdict(list)
for r in seg:
    byq[r[3]].append(r)
main = rows[ad[-2]][3]  # the queue the norm pass runs on is the data-path stream
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
# tail: who finishes last before the optimizer, and a coarse occupancy timeline of both queues
t0 = seg[0][1]
opt = [r for r in seg if 'sumsq_kernel' in r[0]]
t_opt = opt[0][1] if opt else seg[-1][1]
for q, rs in byq.items():
    pre = [r for r in rs if r[2] <= t_opt]
    if pre:
        print("queue %s: last kernel before the optimizer ends at %.3f ms (%s); optimizer starts at %.3f ms" % (
            q, (pre[-1][2] - t0) / 1e6, pre[-1][0].replace('void ', '').replace('(anonymous namespace)::', '')[:50], (t_opt - t0) / 1e6))
bucket = 0.5e6
nb = int((seg[-1][2] - t0) / bucket) + 1
for q, rs in byq.items():
    occ = [0.0] * nb
    for r in rs:
        s, e = r[1] - t0, r[2] - t0
        b0 = int(s / bucket)
        while s < e:
            be = min(e, (b0 + 1) * bucket)
            occ[b0] += be - s
            s = be
            b0 += 1
    print("queue %s busy %% per 0.5 ms: %s" % (q, " ".join("%3d" % round(100 * o / bucket) for o in occ)))
# largest idle gaps on the main queue (what it waited for)
rs = byq[main]
gaps = sorted(((rs[i + 1][1] - rs[i][2], i) for i in range(len(rs) - 1)), reverse=True)[:8]
for g, i in gaps:
    print("main gap %.1f us at %.3f ms: after %s -> before %s" % (g / 1e3, (rs[i][2] - t0) / 1e6,
          rs[i][0].replace('void ', '').replace('(anonymous namespace)::', '')[:40], rs[i + 1][0].replace('void ', '').replace('(anonymous namespace)::', '')[:40]))
