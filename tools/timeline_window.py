"""Print every dispatch of a `bench.py --timeline FILE` record between two instants (ms from the step's first kernel),
both streams side by side in start order.   usage: python tools/timeline_window.py FILE T0_MS T1_MS"""
import sys

rows = []
for ln in open(sys.argv[1]):
    tag, st, t0, d = ln.rstrip("\n").split("\t")
    rows.append((float(t0), float(d), int(st), tag))
t_begin = min(r[0] for r in rows)
lo, hi = float(sys.argv[2]) * 1e3, float(sys.argv[3]) * 1e3
streams = sorted({r[2] for r in rows})
for t0, d, st, tag in sorted(rows):
    a = t0 - t_begin
    if a + d < lo or a > hi:
        continue
    print("%9.1f us  +%7.1f  %s%s" % (a, d, "    " * streams.index(st) + ("[%d] " % st), tag[:110]))
