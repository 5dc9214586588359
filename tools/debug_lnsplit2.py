"""GPU debugging aid, second step: which of the two layer-norm backward schedules leaves the emulator's numbers on
hardware?  `emu` mode (CPU container) stores the micro configuration's gradients for both schedules under tools/_dbg/;
`hip` mode (GPU box) compares the hardware gradients of each schedule with both stored sets."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest  # noqa: E402
import test_engine as TE  # noqa: E402

DBG = os.path.join(os.path.dirname(__file__), "_dbg")


def run(backend, cfg, env):
    os.environ.pop("S2ST_LN_BWD_SPLIT", None)
    os.environ.update(env)
    D = importlib.import_module(TE.DATA)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    a, e = TE.make_engine(backend, cfg, precise=False)
    e.forward(s, training=True, seed=9)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    return {n: gv.detach().cpu().clone() for n, pv, gv, isb in e.named_views() if not isb}


def main():
    kind = sys.argv[1]
    backend = conftest.Backend(kind)
    cfg = dict(TE.MICRO_POSTLN, dropout=0.1, attention_dropout=0.1, activation_dropout=0.05, prenet_dropout=0.5,
               postnet_dropout=0.5)
    fused = run(backend, cfg, {})
    split = run(backend, cfg, {"S2ST_LN_BWD_SPLIT": "1"})
    if kind == "emu":
        os.makedirs(DBG, exist_ok=True)
        torch.save({"fused": fused, "split": split}, os.path.join(DBG, "emu_grads.pt"))
        print("emulator: fused vs split max rel", max(float((fused[n] - split[n]).norm() / (fused[n].norm() + 1e-30)) for n in fused))
        return
    ref = torch.load(os.path.join(DBG, "emu_grads.pt"))
    rows = []
    for n in fused:
        nr = float(ref["fused"][n].norm()) + 1e-30
        rows.append((float((fused[n] - ref["fused"][n]).norm()) / nr, float((split[n] - ref["split"][n]).norm()) / nr,
                     float((fused[n] - split[n]).norm()) / nr, nr, n))
    rows.sort(key=lambda r: -max(r[0], r[1]))
    print("%-12s %-12s %-12s %-10s name" % ("hwF-emuF", "hwS-emuS", "hwF-hwS", "norm"))
    for r in rows[:25]:
        print("%-12.2e %-12.2e %-12.2e %-10.2e %s" % r)


main()
