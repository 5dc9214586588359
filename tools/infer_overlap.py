#!/usr/bin/env python3
"""Do the deferred vocoder's launches overlap the next batch's decoding steps?  From a rocprofv3 kernel trace of
`bench.py --config infer_base`: per queue, busy time and first / last kernel; for the Griffin-Lim kernels how much of their
span falls inside the span of decode-step kernels on another queue, and the decode kernels' mean duration while a
Griffin-Lim kernel is running vs while none is.  usage: infer_overlap.py run_results.db"""
import sqlite3
import sys
from bisect import bisect_right

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
t0 = rows[0][1]
gl = [(s, e) for n, s, e, q in rows if "gl_istft_ola" in n or "gl_stft_project" in n]
dec = [(n, s, e) for n, s, e, q in rows if "gemm_skinny" in n or "decode_attn" in n]
queues = {}
for n, s, e, q in rows:
    d = queues.setdefault(q, [0, 0.0, s, e])
    d[0] += 1
    d[1] += e - s
    d[3] = max(d[3], e)
for q, (n, busy, s, e) in queues.items():
    print("queue %s: %d kernels, busy %.1f ms, active %.1f .. %.1f ms" % (q, n, busy / 1e6, (s - t0) / 1e6, (e - t0) / 1e6))
gl_starts = [s for s, e in gl]
inside = {True: [0, 0.0], False: [0, 0.0]}
for n, s, e in dec:
    i = bisect_right(gl_starts, s) - 1
    during = i >= 0 and gl[i][1] > s
    inside[during][0] += 1
    inside[during][1] += e - s
for k in (False, True):
    c, t = inside[k]
    print("decode kernels %s a Griffin-Lim kernel: %d launches, mean %.2f us" % ("DURING" if k else "outside", c, t / max(c, 1) / 1e3))
tot_gl = sum(e - s for s, e in gl)
print("Griffin-Lim kernels: %d launches, %.1f ms; mean %.1f us" % (len(gl), tot_gl / 1e6, tot_gl / max(len(gl), 1) / 1e3))
# coarse timeline over the timed window of the first vocoder queue: busy % per 5 ms bucket of every queue, and the names
# of the kernels that START on the main queue in each bucket
qs = sorted(queues)
voc_q = [q for q in qs if queues[q][0] < 20000]
if voc_q:
    w0, w1 = queues[voc_q[0]][2], queues[voc_q[0]][3]
    bucket = 5e6
    nb = int((w1 - w0) / bucket) + 1
    occ = {q: [0.0] * nb for q in qs}
    for n, s, e, q in rows:
        if e < w0 or s > w1:
            continue
        s, e = max(s, w0) - w0, min(e, w1) - w0
        b = int(s / bucket)
        while s < e and b < nb:
            be = min(e, (b + 1) * bucket)
            occ[q][b] += be - s
            s = be
            b += 1
    for q in qs:
        print("queue %s busy %% per 5 ms: %s" % (q, " ".join("%3d" % round(100 * o / bucket) for o in occ[q])))
