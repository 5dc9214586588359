"""Pairwise queue-sharing matrix of N torch streams (warm), and how many run at once: python tools/r05_queue_matrix.py [N]"""
import sys, time, torch
d = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CH, SP = 40, 50000


def run(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            for _ in range(CH):
                torch.cuda._sleep(SP)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


ss = [torch.cuda.current_stream(d)] + [torch.cuda.Stream(d) for _ in range(N)]
for s in ss:
    run([s])
one = min(run([ss[1]]) for _ in range(3))
print("one chain %.2f ms; stream 0 = the caller's" % one)
for i in range(len(ss)):
    row = ""
    for j in range(len(ss)):
        if j <= i:
            row += "  . "
        else:
            t = min(run([ss[i], ss[j]]), run([ss[i], ss[j]]))
            row += "  X " if t > 1.6 * one else "  - "
    print("%2d %s" % (i, row))
for n in range(2, len(ss) + 1):
    print("streams 0..%d together: %.2f ms (%.1f x one chain)" % (n - 1, run(ss[:n]), run(ss[:n]) / one))
