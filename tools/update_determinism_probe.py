"""Single process, bf16 mode, micro model: the same three updates ([b0, b1], [b0, b1], [b0]) replayed from the same
state; prints the gradient norm of each update and a parameter checksum after it, per repetition.
usage: python tools/update_determinism_probe.py"""
import importlib, os, sys
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
p0 = eng.params.clone(); buf0 = eng.buffers.clone()
plan = [[b0, b1], [b0, b1], [b0]] if len(sys.argv) < 2 else [[b0]] * 3
for rep in range(int(os.environ.get('REPS', '12'))):
    eng.params.copy_(p0); eng.buffers.copy_(buf0)
    trainer.exp_avg.zero_(); trainer.exp_avg_sq.zero_(); trainer.num_updates = 0
    eng.forward(b0, training=False)  # (uses up the optimizer's "bf16 copy is fresh" mark: the next forward re-casts the parameters)
    model.set_num_updates(0)
    torch.cuda.synchronize()
    out = []
    for u, bs in enumerate(plan):
        r = trainer.train_step(list(bs))
        if not os.environ.get("NOSYNC"): torch.cuda.synchronize()
        out.append("%.7f/%.6f" % (float(r["gnorm"]), float(eng.params.double().abs().sum())))
    print("rep %2d  gnorm/param-checksum per update: %s" % (rep, "  ".join(out)), flush=True)
