#!/usr/bin/env python3
"""Round 5 probe (timing only): what the GPU can do with two HALF-batch training steps in flight at once when the host is
out of the picture -- each trainer's whole step (forward, backward on two streams, norm, Adam) captured ONCE into a
hipGraph on a fixed batch and replayed; graph launches cost ~15 us, so unlike tools/r05_two_engines.py (two host threads
enqueueing 2 x 630 launches through one runtime: ~7 ms of host time per pair) the replay is GPU-bound.  Also: the full
step as a graph against the same step enqueued eagerly (does a graph shorten the gaps between dependent kernels?).
Replays repeat one batch with one seed: timing only."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import s2st_amd  # noqa: E402,F401
PKG = "speech-to-speech-translation_amd"
C_ = importlib.import_module(PKG + ".configs")
tasks = importlib.import_module(PKG + ".tasks")
trainer_mod = importlib.import_module(PKG + ".trainer")


def build(max_tokens, which, dev):
    a = C_.recipe_args("base_recipe")
    task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
    torch.manual_seed(1)
    model = task.build_model(a)
    tr = trainer_mod.Trainer(a, task, model, task.build_criterion(a))
    corpus = task.load_dataset("train", n_utts=4096, seed=1234, with_audio=False)
    batches = corpus.batches(max_tokens=max_tokens, bsz_mult=8)
    order = np.random.RandomState(7).permutation(len(batches))
    s = corpus.collate_batch(batches[order[which]])
    p = model.prepare_sample(s, training=True)
    tr.engine.reserve([p])
    return tr, p, a.n_frames_per_step * s["ntokens"]


def eager(tr, p, n):
    for _ in range(5):
        tr.train_step([p])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step([p])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def capture(tr, p):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tr.train_step([p])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        tr.train_step([p])
    torch.cuda.synchronize()
    return g


def replay(graphs, n):
    streams = [torch.cuda.Stream() for _ in graphs]
    for g, s in zip(graphs, streams):
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    full, pf, ff = build(20000, 0, dev)
    print(f"full batch ({ff} mel frames), eager: {eager(full, pf, n):.3f} ms/step")
    try:
        gf = capture(full, pf)
        print(f"full batch, one hipGraph per step: {replay([gf], n):.3f} ms/step")
    except Exception as e:  # noqa: BLE001
        print("capture of the full step failed:", repr(e)[:300])
        return
    ha, pa, fa = build(10000, 0, dev)
    hb, pb, fb = build(10000, 1, dev)
    print(f"half batch A ({fa} mel frames), eager: {eager(ha, pa, n):.3f} ms/step")
    ga, gb = capture(ha, pa), capture(hb, pb)
    print(f"half batch A, graph alone: {replay([ga], n):.3f} ms/step;  B ({fb} frames) alone: {replay([gb], n):.3f}")
    for _ in range(2):
        print(f"half batches A and B as two graphs in flight at once: {replay([ga, gb], n):.3f} ms per PAIR "
              f"({fa + fb} mel frames; each graph holds its own optimizer update, ~0.45 ms alone)")
    print(f"... and the full-batch graph beside nothing again: {replay([gf], n):.3f} ms/step")
    qa, pqa, fqa = build(6700, 0, dev)
    qb, pqb, fqb = build(6700, 1, dev)
    qc, pqc, fqc = build(6700, 2, dev)
    gs = [capture(qa, pqa), capture(qb, pqb), capture(qc, pqc)]
    print(f"three third-size batches ({fqa + fqb + fqc} frames) as three graphs at once: {replay(gs, n):.3f} ms per TRIPLE; "
          f"one alone {replay(gs[:1], n):.3f}")


if __name__ == "__main__":
    main()
