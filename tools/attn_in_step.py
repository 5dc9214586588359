"""The fused attention kernels INSIDE the training step, by batch geometry: one forward + backward of the engine per batch of
the bench corpus (max-tokens 20000), per-dispatch events on; average launch time of flash_fwd / flash_bwd and the step's
kernel time.   python tools/attn_in_step.py"""
import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import s2st_amd  # noqa
PKG = "speech-to-speech-translation_amd"
C_ = importlib.import_module(PKG + ".configs")
tasks = importlib.import_module(PKG + ".tasks")
bd = importlib.import_module(PKG + ".runtime.binding")
dev = torch.device("cuda:0")
a = C_.recipe_args("base_recipe")
task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
torch.manual_seed(1)
model = task.build_model(a)
eng = model.engine
corpus = task.load_dataset("train", n_utts=4096, seed=1234)
batches = corpus.batches(max_tokens=20000, bsz_mult=8)
lib = bd.lib()
lib.s2st_profile_enable.argtypes = [C.c_int32]
lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
lib.s2st_profile_report.restype = C.c_int64
sel = sorted(range(len(batches)), key=lambda i: -len(batches[i]))
pick = [sel[0], sel[len(sel) // 6], sel[len(sel) // 3], sel[len(sel) // 2], sel[2 * len(sel) // 3], sel[5 * len(sel) // 6], sel[-1]]
prepared = [model.prepare_sample(corpus.collate_batch(batches[i]), training=True) for i in pick]
eng.reserve(prepared)
print("utts  E    D    flash_fwd us (n)   flash_bwd us (n)   attention ms   all kernels ms")
for i, p in zip(pick, prepared):
    for rep in range(3):
        if rep == 2:
            lib.s2st_profile_enable(1)
        eng.forward(p, training=True, seed=5)
        eng.zero_grad()
        eng.backward(1.0)
        torch.cuda.synchronize()
    lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 17)
    n = lib.s2st_profile_report(buf, len(buf))
    tot, fw, bw = 0.0, (0.0, 0), (0.0, 0)
    tags = {}
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        f = line.split("\t")
        tot += float(f[2])
        if f[0].startswith("flash_"):
            tags[f[0]] = (float(f[2]) / max(int(f[1]), 1), int(f[1]))
        if f[0].startswith("flash_fwd"):
            fw = (fw[0] + float(f[2]), fw[1] + int(f[1]))
        if f[0].startswith("flash_bwd"):
            bw = (bw[0] + float(f[2]), bw[1] + int(f[1]))
    B = len(batches[i])
    E = int(p.batch.E) if hasattr(p, "batch") else -1
    D = int(p.batch.D) if hasattr(p, "batch") else -1
    print("%4d %4d %4d   %8.1f (%2d)      %8.1f (%2d)      %8.3f      %8.3f" % (B, E, D, fw[0] / max(fw[1], 1), fw[1], bw[0] / max(bw[1], 1), bw[1],
                                                                          (fw[0] + bw[0]) / 1e3, tot / 1e3),
          "  ".join("%s %.1f us x %d" % (k, v[0], v[1]) for k, v in sorted(tags.items())))
