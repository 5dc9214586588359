"""GPU debugging aid, third step: S2ST_DEBUG_LN_DUMP writes the inputs and outputs of every layer-norm backward call; the
two schedules' dumps are compared call by call (the first array that differs is where they part ways)."""
import glob
import importlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    sys.path.insert(0, os.path.join(HERE, ".."))
    import conftest
    import test_engine as TE
    backend = conftest.Backend(sys.argv[2])
    cfg = dict(TE.MICRO_POSTLN, dropout=0.1, attention_dropout=0.1, activation_dropout=0.05, prenet_dropout=0.5,
               postnet_dropout=0.5)
    D = importlib.import_module(TE.DATA)
    c = D.SyntheticFisherCorpus(n_utts=4, seed=3, max_src=64, median_src=50, min_src=30)
    s = c.collate_batch(range(4))
    a, e = TE.make_engine(backend, cfg, precise=False)
    e.forward(s, training=True, seed=9)
    e.zero_grad()
    e.backward(1.0)
    backend.sync()
    sys.exit(0)

kind = sys.argv[1] if len(sys.argv) > 1 else "hip"
dirs = {}
for mode, env in (("fused", {}), ("split", {"S2ST_LN_BWD_SPLIT": "1"})):
    d = f"/tmp/lndump_{mode}"
    os.makedirs(d, exist_ok=True)
    for f in glob.glob(d + "/*.bin"):
        os.remove(f)
    subprocess.run([sys.executable, __file__, "child", kind], env=dict(os.environ, S2ST_DEBUG_LN_DUMP=d, **env), check=True,
                   stderr=subprocess.DEVNULL if mode == "split" else None)
    dirs[mode] = d
for f in sorted(glob.glob(dirs["fused"] + "/*.bin")):
    g = f.replace(dirs["fused"], dirs["split"])
    a, b = np.fromfile(f, dtype=np.float32), np.fromfile(g, dtype=np.float32)
    nd = int((a != b).sum())
    if nd:
        idx = np.nonzero(a != b)[0]
        print(f"{os.path.basename(f):22s} differs in {nd} of {a.size} (first at {idx[:6]}), max abs {np.abs(a - b).max():.3e} rel {np.abs(a - b).max() / (np.abs(a).max() + 1e-30):.2e}")
    else:
        print(f"{os.path.basename(f):22s} equal ({a.size})")
