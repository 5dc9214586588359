""""top data-path kernels: alone us / in-step us / launches".  The same 20 timed batches twice through
bench.py's per-dispatch replay (S2ST_BENCH_VERBOSE=1 prints every kernel tag's launches per step and average duration):
  * S2ST_NO_SIDE_STREAM=1 -- everything on ONE stream: a kernel runs with the chip to itself ("alone": same shapes, same data,
    same caches as in the step -- not a micro-benchmark on hot operands);
  * the default schedule -- weight gradients, folds, aux heads on the second stream beside the data path ("in-step").
usage: python tools/alone_vs_instep.py [extra bench args]        (GPU box)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env):
    e = dict(os.environ, S2ST_BENCH_VERBOSE="1", **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-other-configs",
                        "--cpu-seconds", "0", "--no-host-fed"] + sys.argv[1:], env=e, capture_output=True, text=True)
    rows, ms = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"\]\s+(\S.*?)\s+launches/step\s+([\d.]+)\s+avg\s+([\d.]+) us\s+ms/step\s+([\d.]+)", ln)
        if m:
            rows[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            import json
            ms = json.loads(ln)["ms_per_step"]
    return rows, ms


alone, ms_a = run({"S2ST_NO_SIDE_STREAM": "1"})
step, ms_s = run({})
print("# ms per step: one stream %.3f, default schedule %.3f" % (ms_a, ms_s))
print("# %-62s %9s %9s %9s %7s %9s" % ("kernel (tag = name in a rocprofv3 trace)", "launches", "alone us", "in-step", "ratio", "ms/step"))
tot_a = tot_s = 0.0
for tag, (n, us, msk) in sorted(step.items(), key=lambda kv: -kv[1][2])[:16]:
    a = alone.get(tag)
    print("%-64s %9.1f %9.2f %9.2f %7.2f %9.3f" % (tag[:64], n, a[1] if a else float("nan"), us, us / a[1] if a else float("nan"), msk))
for tag, (n, us, msk) in step.items():
    tot_s += msk
    if tag in alone:
        tot_a += alone[tag][2]
print("# all profiled kernels: %.3f ms per step alone, %.3f ms per step in-step (two streams overlap: the step takes %.3f)" % (tot_a, tot_s, ms_s))
