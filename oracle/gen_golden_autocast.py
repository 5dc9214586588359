#!/usr/bin/env python3
"""What does the REFERENCE's own mixed precision do to the gradients?  (build container only)
    python oracle/gen_golden_autocast.py      # writes tests/golden/s2st_base_autocast.npz
TEST INFRASTRUCTURE.  The reference model + criterion (base geometry, the seeded 8-utterance base batch, dropouts 0,
name-keyed synthetic weights: the set-up of tests/golden/s2st_base.npz) run once under ``torch.autocast("cpu",
dtype=torch.bfloat16)`` -- PyTorch's mixed-precision policy applied to the reference's code: linear / conv / matmul on
bf16 operands AND bf16 results, layer norm / softmax / losses in fp32 -- and every gradient tensor is compared with the
fp32 golden's sample of it (``gsub.*``) by the measure tests/test_engine.py::check_gradient_direction uses.  The per-tensor
errors are the yardstick for the HIP path's benchmarked bf16 mode: it keeps fp32 results and a fp32
residual stream, so it has to stay within 1.5 x of these."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys_args = sys.argv[1:]
sys.argv = [sys.argv[0]]
import gen_golden as GG  # noqa: E402
from configs import CONFIGS  # noqa: E402
from synth_weights import load_synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    which = sys_args or ["base", "hubert_train"]
    if "base" in which:
        z = np.load(os.path.join(OUT, "s2st_base.npz"))
        torch.manual_seed(0)
        a, model, crit = GG.build_reference(CONFIGS["base"])
        load_synth(model, seed=0)
        record(z, model, crit, GG.sample_for("base", 0), "s2st_base_autocast.npz")
    if "hubert_train" in which:
        # config 3 / 4: the frozen HuBERT front end (reference HubertModel, hubert_base geometry) + base model + aux heads on
        # the 4-utterance batch of tests/golden/s2st_hubert_train.npz -- the front end runs under autocast too
        import gen_golden_hubert_train as GT
        from configs import hubert_train_sample
        z = np.load(os.path.join(OUT, "s2st_hubert_train.npz"))
        cfg = CONFIGS["hubert_train"]
        hub = GT.GH.build(GT.HO.HUBERT_CONFIGS[cfg.get("hubert_geometry_name", "base")])
        a, model, crit = GT.build(cfg, hub)
        record(z, model, crit, hubert_train_sample(0), "s2st_hubert_train_autocast.npz")


def record(z, model, crit, sample, fname):
    model.train()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss, ss, log = crit(model, sample)
    loss.backward()
    named = dict(model.named_parameters())
    names = [k[5:] for k in z.files if k.startswith("gsub.")]
    gmax = max(float(np.linalg.norm(z["gsub." + n])) for n in names)
    errs, num, den = [], 0.0, 0.0
    for n in names:
        ref = z["gsub." + n].astype(np.float64).reshape(-1)
        mine = GG.gsub(GG.to_np(named[n].grad.float())).astype(np.float64).reshape(-1)
        d, r = float(np.linalg.norm(mine - ref)), float(np.linalg.norm(ref))
        num += d * d
        den += r * r
        errs.append(d / (r + 1e-3 * gmax))
    whole = float(np.sqrt(num / den))
    rec = {"names": np.asarray(names), "err": np.asarray(errs), "whole": np.asarray(whole),
           "loss_autocast": np.asarray(float(loss)), "loss_fp32": z["log.loss"]}
    for k, v in log.items():
        rec[f"log.{k}"] = np.asarray(float(v))
    np.savez_compressed(os.path.join(OUT, fname), **rec)
    order = np.argsort(errs)[::-1]
    print(fname, "autocast(bf16) vs fp32 golden: loss %.5f vs %.5f; whole gradient %.3e; worst tensors:" % (
        float(loss), float(z["log.loss"]), whole))
    for i in order[:10]:
        print("   %-70s %.3e" % (names[i], errs[i]))
    print("   median %.3e" % float(np.median(errs)))


if __name__ == "__main__":
    main()
