"""Which HIP streams share a hardware queue?  Pairs of streams each run a dependent chain of ~25 us spin kernels at once; a pair
that serialises (time ~ 2 x one chain) shares a queue.  Streams are made in creation order at the priorities given.
python tools/r05_queue_probe.py"""
import time, torch
d = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range())
lo, hi = torch.cuda.Stream.priority_range()
N = 100
SLEEP = 60000  # cycles per kernel (torch.cuda._sleep: a spin kernel; ~25 us -- the host enqueues a launch in ~5)


def chain(s, x):
    with torch.cuda.stream(s):
        for _ in range(N):
            torch.cuda._sleep(SLEEP)


def run(streams):
    xs = [torch.zeros(64, device=d) for _ in streams]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, x in zip(streams, xs):
        chain(s, x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for label, prios in (("8 x normal", [0] * 8), ("normal / high alternating", [0, hi, 0, hi, 0, hi, 0, hi]),
                     ("low..high spread", list(range(hi, lo + 1)) * 3)):
    ss = [torch.cuda.Stream(d, priority=p) for p in prios]
    for s_ in ss:  # (first use of a stream creates its hardware queue: milliseconds -- keep it out of the timings)
        run([s_])
    for _ in range(2):
        run(ss[:1])
    one = run(ss[:1])
    print("==", label, prios, "one chain %.2f ms" % one)
    for i in range(1, len(ss)):
        t = run([ss[0], ss[i]])
        print("   stream 0 + stream %d (prio %d): %.2f ms %s" % (i, prios[i], t, "<- serialised" if t > 1.6 * one else ""))
    for n in (3, 4, 6, 8):
        if n <= len(ss):
            print("   first %d together: %.2f ms" % (n, run(ss[:n])))
