#!/usr/bin/env python3
"""Config 5 timing on the GPU: AR mel decode (key/value caches) + Griffin-Lim (64 iterations) on
Fisher-shaped inputs, base geometry, random-init weights.  python tools/infer_bench.py [B] [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import s2st_amd  # noqa
import s2st_oracle as O
from configs import CONFIGS
PKG = "speech-to-speech-translation_amd"
tasks = importlib.import_module(PKG + ".tasks")
G = importlib.import_module(PKG + ".speech_generator")
V = importlib.import_module(PKG + ".vocoder")
D = importlib.import_module(PKG + ".data")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
a = O.make_args(**CONFIGS["base_recipe"])
task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
model = task.build_model(a)
corpus = D.SyntheticFisherCorpus(n_utts=256, seed=9)
s = corpus.collate_batch(range(B))
s["net_input"]["collated_audios_orig"] = None
s["net_input"]["padding_mask"] = None
voc = V.GriffinLimVocoder(sample_rate=24000, win_size=1200, hop_size=300, n_fft=2048, n_mels=80, f_min=20, f_max=8000,
                          spec_bwd_max_iter=64, device=dev)
gen = G.AutoRegressiveSpeechGenerator(model, None, None, max_iter=steps, eos_prob_threshold=2.0)  # fixed work: never stops early
gen.generate(model, s)
torch.cuda.synchronize()
t = time.perf_counter()
fin = gen.generate(model, s)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"AR decode: B={B} x {steps} steps ({steps*4} mel frames each): {dt*1e3:.1f} ms -> {dt/steps*1e3:.3f} ms/step, {B/dt:.1f} utt/s", flush=True)
feat = fin[0]["feature"]
voc(feat)
torch.cuda.synchronize()
t = time.perf_counter()
for b in range(min(B, 4)):
    w = voc(fin[b]["feature"])
torch.cuda.synchronize()
dv = (time.perf_counter() - t) / min(B, 4)
print(f"Griffin-Lim 64 iters on {feat.shape[0]} frames -> {w.shape[1]} samples: {dv*1e3:.1f} ms/utt ({w.shape[1]/24000/dv:.0f}x real time)")
print(f"end-to-end (per-utterance vocoder): {B/(dt + dv*B):.2f} utt/s")
feats = [f["feature"] for f in fin]
voc.batch(feats)
torch.cuda.synchronize()
t = time.perf_counter()
ws = voc.batch(feats)
torch.cuda.synchronize()
db = time.perf_counter() - t
print(f"Griffin-Lim batched over {B} utterances: {db*1e3:.1f} ms ({db/B*1e3:.2f} ms/utt)")
print(f"end-to-end (batched vocoder): {B/(dt + db):.2f} utt/s")
