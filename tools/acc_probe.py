import os, sys, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import torch
import s2st_amd
import s2st_oracle as O
from synth_weights import load_synth
from test_engine import MICRO, nano_batches
PKG = "speech-to-speech-translation_amd"
bd = importlib.import_module(PKG + ".runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
tasks = importlib.import_module(PKG + ".tasks")
cfg = dict(MICRO, encoder_embed_dim=128, decoder_embed_dim=128, encoder_attention_heads=2, decoder_attention_heads=2)
a = O.make_args(**cfg)
a.precise_gemm = False
task = tasks.S2ST_TranslationTask.setup_task(a, device=torch.device("cuda", 0))
model = task.build_model(a)
load_synth(model, 0)
e = model.engine
b0, b1 = nano_batches()
def grads_of(bs, seeds, zero_each):
    e.zero_grad()
    out = []
    for s, sd in zip(bs, seeds):
        if zero_each: e.zero_grad()
        e.forward(s, training=True, seed=sd)
        e.backward(1.0)
        torch.cuda.synchronize()
        if zero_each: out.append(e.grads.clone())
    return out if zero_each else e.grads.clone()
for rep in range(4):
    g0, g1 = grads_of([b0, b1], [5, 6], True)
    acc = grads_of([b0, b1], [5, 6], False)
    ref = g0 + g1
    d = (acc - ref)
    print("rep", rep, "rel", float(d.norm() / ref.norm()), "|acc|", float(acc.norm()), "|ref|", float(ref.norm()))
    if float(d.norm() / ref.norm()) > 1e-5:
        # per tensor
        for n, off, num, shp in e.param_infos if hasattr(e, "param_infos") else []:
            dd = d[off:off + num].norm(); rr = ref[off:off + num].norm()
            if float(dd) > 1e-4 * float(rr) + 1e-7: print("   ", n, float(dd), float(rr))
