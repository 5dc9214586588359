#!/usr/bin/env python3
"""Round 5 probe (timing only, not a product path): an UPPER bound of what a two-chain schedule of the training step could
cost -- two complete Trainer objects in ONE process, each stepping half-size batches (max-tokens 10000) on its own stream
from its own host thread (ctypes drops the GIL), against one Trainer stepping the full batches (max-tokens 20000).
Each trainer runs its own optimizer update (0.45 ms that a real two-chain step would run once) and its own post-net /
front end, and nothing is shared; BatchNorm does not couple the halves (results differ from a one-batch step: timing only).
Usage: python tools/r05_two_engines.py [steps]"""
import importlib
import os
import sys
import threading
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


def build(max_tokens, need, dev):
    a = C_.recipe_args("base_recipe")
    task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
    torch.manual_seed(1)
    model = task.build_model(a)
    tr = trainer_mod.Trainer(a, task, model, task.build_criterion(a))
    corpus = task.load_dataset("train", n_utts=4096, seed=1234, with_audio=False)
    batches = corpus.batches(max_tokens=max_tokens, bsz_mult=8)
    order = np.random.RandomState(7).permutation(len(batches))
    mine = [batches[order[i % len(batches)]] for i in range(need)]
    prepared = [model.prepare_sample(corpus.collate_batch(ix), training=True) for ix in mine]
    tr.engine.reserve(prepared)
    frames = sum(a.n_frames_per_step * corpus.collate_batch(ix)["ntokens"] for ix in mine[5:])
    return tr, prepared, frames


def run(trs, steps, warm=5):
    """every (trainer, batches, stream) steps `steps` times on its own thread; returns wall seconds of the timed part"""
    bar = threading.Barrier(len(trs) + 1)
    issue = []

    def work(tr, prepared, stream):
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            for i in range(warm):
                tr.train_step([prepared[i]])
            stream.synchronize()
            bar.wait()
            t0 = time.perf_counter()
            for i in range(warm, warm + steps):
                tr.train_step([prepared[i]])
            issue.append((time.perf_counter() - t0) / steps * 1e3)  # host time to ENQUEUE a step (queue back-pressure included)
            stream.synchronize()
        bar.wait()

    th = [threading.Thread(target=work, args=t) for t in trs]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    print("   host enqueue per step, per thread (ms):", ", ".join(f"{x:.2f}" for x in issue))
    return dt


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    full, pf, ff = build(20000, steps + 5, dev)
    dt = run([(full, pf, torch.cuda.Stream())], steps)
    print(f"one trainer, max-tokens 20000: {dt / steps * 1e3:.3f} ms/step, {ff / dt / 1e3:.0f} k mel-frames/s")
    del full, pf
    a_, pa, fa = build(10000, steps + 5, dev)
    dt = run([(a_, pa, torch.cuda.Stream())], steps)
    print(f"one trainer, max-tokens 10000: {dt / steps * 1e3:.3f} ms/step, {fa / dt / 1e3:.0f} k mel-frames/s")
    b_, pb, fb = build(10000, steps + 5, dev)
    pb = pb[1:] + pb[:1]  # (other batches than trainer A at the same time)
    run([(b_, pb, torch.cuda.Stream())], 20)  # (first-use costs of the second engine out of the way)
    for rep in range(2):
        dt = run([(a_, pa, torch.cuda.Stream()), (b_, pb, torch.cuda.Stream())], steps)
        print(f"two trainers at once, max-tokens 10000 each: {dt / steps * 1e3:.3f} ms per PAIR of half-steps, "
              f"{(fa + fb) / dt / 1e3:.0f} k mel-frames/s (each pair runs TWO optimizer updates, ~0.45 ms each alone)")


if __name__ == "__main__":
    main()
