"""Cold process: the three updates of the micro model ([b0, b1], [b0, b1], [b0]); after each, position-weighted checksums
(sign- and order-sensitive) of every gradient and parameter tensor -- to compare between processes (tools/cold_grad_diff.sh)."""
import importlib, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import s2st_amd, s2st_oracle as O
from synth_weights import load_synth
from test_engine import MICRO, nano_batches
PKG = "speech-to-speech-translation_amd"
bd = importlib.import_module(PKG + ".runtime.binding"); bd.load_library(bd.DEFAULT_LIB, emulator=False)
tasks = importlib.import_module(PKG + ".tasks"); tr = importlib.import_module(PKG + ".trainer")
cfg = dict(MICRO, encoder_embed_dim=128, decoder_embed_dim=128, encoder_attention_heads=2, decoder_attention_heads=2)
a = O.make_args(**cfg); a.precise_gemm, a.lr, a.warmup_updates, a.clip_norm = False, 1e-3, 1, 0.05
task = tasks.S2ST_TranslationTask.setup_task(a, device=torch.device("cuda", 0))
model = task.build_model(a); load_synth(model, 0)
trainer = tr.Trainer(a, task, model, task.build_criterion(a))
eng = model.engine
b0, b1 = nano_batches()
out = {"steps": []}
def chk(v):
    w = torch.cos(torch.arange(v.numel(), dtype=torch.float64) * 0.37 + 1.0)
    return [float((v * w).sum()), float(v.abs().sum())]
for u, bs in enumerate([[b0, b1], [b0, b1], [b0]]):
    r = trainer.train_step(list(bs)); torch.cuda.synchronize()
    g = eng.grads.double().cpu(); p = eng.params.double().cpu()
    rec = {"gnorm": float(r["gnorm"]), "stats": [float(x) for x in trainer.criterion.last_outputs["stats"].double().cpu()[16:24]], "t": {}}
    for n, t in model.named_parameters():
        off = (t.data_ptr() - eng.params.data_ptr()) // 4
        rec["t"][n] = chk(g[off:off + t.numel()]) + chk(p[off:off + t.numel()])
    out["steps"].append(rec)
json.dump(out, open(sys.argv[1], "w"))
print("gnorm3 %.7f" % out["steps"][2]["gnorm"])
