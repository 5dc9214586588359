"""Two ranks (bf16 mode, one GPU, gloo) vs one process accumulating both batches: gradient norms of the three updates and
the parameters that differ most -- what the flaky check of tests/test_distributed.py sees.  usage: python tools/two_rank_gnorm_probe.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.multiprocessing as mp
import test_distributed as T

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 2000
    procs = [ctx.Process(target=T._worker, args=(r, 2, port, q, True, True)) for r in range(2)]
    for p in procs: p.start()
    params = q.get(timeout=300)
    gn = q.get(timeout=60)
    for p in procs: p.join(timeout=120)
    import s2st_amd, s2st_oracle as O
    from synth_weights import load_synth
    from test_engine import MICRO, nano_batches
    PKG = T.PKG
    bd = importlib.import_module(PKG + ".runtime.binding"); bd.load_library(bd.DEFAULT_LIB, emulator=False)
    tasks = importlib.import_module(PKG + ".tasks"); tr = importlib.import_module(PKG + ".trainer")
    cfg = dict(MICRO, encoder_embed_dim=128, decoder_embed_dim=128, encoder_attention_heads=2, decoder_attention_heads=2)
    a = O.make_args(**cfg); a.precise_gemm, a.lr, a.warmup_updates, a.clip_norm = False, 1e-3, 1, 0.05
    task = tasks.S2ST_TranslationTask.setup_task(a, device=torch.device("cuda", 0))
    model = task.build_model(a); load_synth(model, 0)
    trainer = tr.Trainer(a, task, model, task.build_criterion(a))
    b0, b1 = nano_batches()
    ref = []
    for u in range(3):
        r = trainer.train_step([b0, b1] if u < 2 else [b0]); ref.append(float(r["gnorm"]))
    torch.cuda.synchronize()
    print("gnorm two ranks:", ["%.7f" % x for x in gn], " one process:", ["%.7f" % x for x in ref],
          " rel diff:", ["%.1e" % (abs(a_ - b_) / b_) for a_, b_ in zip(gn, ref)])
    worst = sorted(((float((torch.from_numpy(params[n]) - p.detach().cpu()).abs().max()), n) for n, p in model.named_parameters()), reverse=True)[:6]
    print("   largest parameter differences:", [(n, "%.1e" % d) for d, n in worst])
