#!/usr/bin/env python3
"""Golden vectors for ``t2s_transformer`` WITH its feature-level CTC head from the REFERENCE (build container only):
    python oracle/gen_golden_t2s_ctc.py        # writes tests/golden/s2st_tiny_t2s_ctc.npz
TEST INFRASTRUCTURE.  examples/s2s_trans/models/t2s_transformer.py:168-170, 258 (``ctc_proj = Linear(out_dim, |src_dict|)``
over ``feature_out``) through examples/s2s_trans/criterions/t2s_loss.py:134-144 (``F.ctc_loss(log_softmax(...)^T, src
tokens, decoder-step lengths, text lengths, reduction='mean', zero_infinity=True) * ctc_weight``), the tiny text geometry of
gen_golden_t2s.py with ``--ctc-weight 0.3``; the oracle must reproduce loss terms and gradients before the file is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = [sys.argv[0]]
import gen_golden_t2s as GT  # noqa: E402  (reference import path + shims; builds the reference t2s model)
import gen_golden as GG  # noqa: E402
from configs import CONFIGS, golden_sample  # noqa: E402
from synth_weights import load_synth  # noqa: E402
import s2st_oracle as O  # noqa: E402

CFG = dict(CONFIGS["tiny_t2s"], ctc_weight=0.3)


def main():
    a = O.make_args(**CFG)
    model, task = GT.build(a)
    assert tuple(model.decoder.ctc_proj.weight.shape) == (a.src_vocab_size, 320)
    crit = GT.T2SCriterion(task, False, a.n_frames_per_step, False, 0.4, a.bce_pos_weight, a.ctc_weight)
    sample = dict(golden_sample("tiny", 0), speaker=None)
    loss, ss, log = crit(model, sample)
    loss.backward()
    out = {f"log.{k}": np.asarray(float(v)) for k, v in log.items()}
    assert float(log["ctc_loss"]) > 0
    named = dict(model.named_parameters())
    gn = {n: float(p.grad.norm()) for n, p in named.items() if p.grad is not None}
    out["grad_norm_names"] = np.array(sorted(gn))
    out["grad_norms"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
    for n in sorted(gn):
        out[f"gsub.{n}"] = GG.gsub(GG.to_np(named[n].grad))
    out["sd_names"] = np.array(list(model.state_dict().keys()))
    m = O.S2STModel(O.make_args(**CFG))
    load_synth(m, 0)
    m.train()
    l2, _, lg2, outs = O.criterion_forward(m, sample)
    l2.backward()
    for k in ("loss", "l1_loss", "mse_loss", "eos_loss", "ctc_loss"):
        assert abs(float(lg2[k]) - float(log[k])) < 2e-5 * max(1.0, abs(float(log[k]))), (k, float(lg2[k]), float(log[k]))
    mine = dict(m.named_parameters())
    gmax = max(gn.values())
    for n, p in named.items():
        if p.grad is not None:  # (conv biases in front of a BatchNorm: mathematically zero gradients, rounding noise only)
            d = float((mine[n].grad - p.grad).norm())
            assert d <= 2e-3 * float(p.grad.norm()) + 1e-6 * gmax, (n, d, float(p.grad.norm()))
    path = os.path.join(GG.ROOT, "tests", "golden", "s2st_tiny_t2s_ctc.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: round(float(v), 5) for k, v in log.items()})


if __name__ == "__main__":
    main()
