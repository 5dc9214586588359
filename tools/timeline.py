"""Summarise a per-dispatch timeline written by `bench.py --timeline FILE` (s2st_profile_timeline: tag, stream, start us,
duration us on the GPU clock; kernels of the s2st library only -- memsets / copies / torch's own kernels appear as gaps).

usage: python tools/timeline.py FILE [n_gaps]"""
import sys
from collections import defaultdict


def main():
    rows = []
    for ln in open(sys.argv[1]):
        tag, st, t0, d = ln.rstrip("\n").split("\t")
        rows.append((tag, int(st), float(t0), float(d)))
    ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    t_begin = min(r[2] for r in rows)
    t_end = max(r[2] + r[3] for r in rows)
    print("step span %.3f ms, %d dispatches" % ((t_end - t_begin) * 1e-3, len(rows)))
    by_stream = defaultdict(list)
    for r in rows:
        by_stream[r[1]].append(r)
    for st, rs in sorted(by_stream.items()):
        rs.sort(key=lambda r: r[2])
        busy = sum(r[3] for r in rs)
        first, last = rs[0][2], rs[-1][2] + rs[-1][3]
        gaps = []
        for a, b in zip(rs, rs[1:]):
            g = b[2] - (a[2] + a[3])
            gaps.append((g, a, b))
        small = [g for g, _, _ in gaps if 0 <= g < 20]
        print("stream %d: %d kernels, busy %.3f ms, active span %.3f .. %.3f ms, gaps inside the span %.3f ms "
              "(%d gaps < 20 us: %.3f ms, median %.1f us)" % (
                  st, len(rs), busy * 1e-3, (first - t_begin) * 1e-3, (last - t_begin) * 1e-3,
                  sum(max(g, 0) for g, _, _ in gaps) * 1e-3, len(small), sum(small) * 1e-3,
                  sorted(small)[len(small) // 2] if small else 0.0))
        for g, a, b in sorted(gaps, key=lambda x: -x[0])[:ngaps]:
            print("   gap %8.1f us at %8.3f ms: after %-44s before %s" % (g, (a[2] + a[3] - t_begin) * 1e-3, a[0][:44], b[0][:60]))
        agg = defaultdict(lambda: [0, 0.0])
        for r in rs:
            agg[r[0]][0] += 1
            agg[r[0]][1] += r[3]
        for tag, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("   %-72s %4d  %8.3f ms  %7.1f us" % (tag[:72], n, us * 1e-3, us / n))
    # busy share per 0.5 ms
    w = 500.0
    nb = int((t_end - t_begin) / w) + 1
    for st, rs in sorted(by_stream.items()):
        bins = [0.0] * nb
        for r in rs:
            a, b = r[2] - t_begin, r[2] - t_begin + r[3]
            i = int(a / w)
            while a < b:
                e = min(b, (i + 1) * w)
                bins[i] += e - a
                a = e
                i += 1
        print("stream %d busy %% per 0.5 ms: %s" % (st, " ".join("%3d" % round(100 * x / w) for x in bins)))


if __name__ == "__main__":
    main()
