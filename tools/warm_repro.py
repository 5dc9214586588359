"""One process, bf16 mode (set S2ST_NO_FLASH=1 for the most sensitive schedule): replay [2 updates, then the third step's
forward + backward] from the same state; per repetition, bit-level hashes of the parameters after update 2 and of the third
step's losses / gradients -- which of them repeats, and which tensors do not.  usage: python tools/warm_repro.py [reps]"""
import importlib, os, sys, hashlib
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
names = [(n, (t.data_ptr() - eng.params.data_ptr()) // 4, t.numel()) for n, t in model.named_parameters()]
def hashes(v):
    c = v.cpu().contiguous().numpy().tobytes()
    return {n: hashlib.md5(c[4 * o:4 * (o + k)]).hexdigest()[:8] for n, o, k in names}
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    eng.params.copy_(p0); eng.buffers.copy_(buf0)
    trainer.exp_avg.zero_(); trainer.exp_avg_sq.zero_(); trainer.num_updates = 0; model.set_num_updates(0)
    eng.forward(b0, training=False)
    torch.cuda.synchronize()
    gn = []
    for u in range(2):
        r = trainer.train_step([b0, b1]); torch.cuda.synchronize(); gn.append(float(r["gnorm"]))
    hp = hashes(eng.params)
    hg2 = hashes(eng.grads)
    # third step by hand: forward + backward of b0 with the trainer's seed
    eng.step_seed = (trainer.seed + trainer.num_updates) * 1000003
    eng.zero_grad()
    out = eng.forward(b0, training=True, seed=eng.step_seed)
    stats = out["stats"].clone()
    eng.backward(1.0)
    torch.cuda.synchronize()
    hg3 = hashes(eng.grads)
    cur = dict(hp=hp, hg2=hg2, hg3=hg3, stats=[float(x) for x in stats.double().cpu()[16:23]])
    if ref is None:
        ref = cur
        print("rep 0: reference; losses of step 3:", ["%.7f" % x for x in cur["stats"]])
        continue
    dp = [n for n in hp if hp[n] != ref["hp"][n]]
    dg2 = [n for n in hp if hg2[n] != ref["hg2"][n]]
    dg3 = [n for n in hp if hg3[n] != ref["hg3"][n]]
    print("rep %d: params after update 2 differ (bitwise) in %d tensors %s | gradients of update 2 differ in %d %s | step-3 losses %s | step-3 gradients differ in %d tensors" % (
        rep, len(dp), [n.replace("transformer_layers", "L") for n in dp[:6]], len(dg2), [n.replace("transformer_layers", "L") for n in dg2[:6]],
        "SAME" if cur["stats"] == ref["stats"] else ["%.7f" % x for x in cur["stats"]], len(dg3)))
