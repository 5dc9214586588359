"""GPU debugging aid: the micro post-LN configuration of tests/test_engine.py run (a) twice with defaults, (b) with
S2ST_LN_BWD_SPLIT=1, (c) in bf16x3 mode (same seed, same dropout masks) -- prints, per parameter tensor, how far each of
the bf16 runs is from the other and from the fp32-accurate run."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest  # noqa: E402
import test_engine as TE  # noqa: E402


GLOBAL_ENV = dict(kv.split("=") for kv in sys.argv[2:])


def run(backend, cfg, env, precise=False):
    for k in ("S2ST_LN_BWD_SPLIT",):
        os.environ.pop(k, None)
    os.environ.update(GLOBAL_ENV)
    os.environ.update(env)
    D = importlib.import_module(TE.DATA)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    a, e = TE.make_engine(backend, cfg, precise=precise)
    o = e.forward(s, training=True, seed=9)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    return {n: gv.clone() for n, pv, gv, isb in e.named_views() if not isb}


def main():
    backend = conftest.Backend(sys.argv[1] if len(sys.argv) > 1 else "hip")
    cfg = dict(TE.MICRO_POSTLN, dropout=0.1, attention_dropout=0.1, activation_dropout=0.05, prenet_dropout=0.5,
               postnet_dropout=0.5)
    a0 = run(backend, cfg, {})
    a1 = run(backend, cfg, {})
    b = run(backend, cfg, {"S2ST_LN_BWD_SPLIT": "1"})
    b1 = run(backend, cfg, {"S2ST_LN_BWD_SPLIT": "1"})
    p = run(backend, cfg, {}, precise=True)
    rows = []
    for n in a0:
        nr = float(p[n].norm()) + 1e-30
        rows.append((float((a0[n] - b[n]).norm()) / nr, float((a0[n] - a1[n]).norm()) / nr, float((b[n] - b1[n]).norm()) / nr,
                     float((a0[n] - p[n]).norm()) / nr, float((b[n] - p[n]).norm()) / nr, nr, n))
    rows.sort(reverse=True)
    print("%-10s %-10s %-10s %-10s %-10s %-10s name" % ("fused-split", "fused-rep", "split-rep", "fused-x3", "split-x3", "norm"))
    print('global env', GLOBAL_ENV)
    for r in rows[:int(os.environ.get('DBG_ROWS', 12))]:
        print("%-10.2e %-10.2e %-10.2e %-10.2e %-10.2e %-10.2e %s" % r)


main()
