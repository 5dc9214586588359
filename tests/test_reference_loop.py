"""The plugin driven by the REFERENCE's own training step (`north_star`: "so fairseq_cli.train drives it
unchanged"): `FairseqTask.train_step` + `FairseqAdam` + `clip_grad_norm_` for three updates (and with update-freq 2) on the
plugin's task / model / criterion -- emulator backend, build container -- against the reference's own model and criterion
under the SAME driver.  Proves what INTEGRATION.md claims: gradients arrive in `p.grad` (views of the gradient arena) also
after fairseq's `zero_grad()` has set them to None, gradient accumulation over micro-batches adds, the optimizer's in-place
parameter writes reach the next forward (the bf16 / device copies follow the parameters' version), and the criterion's
logging output carries the reference's keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")
def test_reference_train_step_and_adam_drive_the_plugin():
    subprocess.check_call([os.path.join(ROOT, "tests", "hipemu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_loop_child.py"), ROOT], capture_output=True, text=True,
                       timeout=1500)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(line[0][7:])
    for uf in ("uf1", "uf2"):
        o = out[uf]
        assert o["names_equal"]
        for a, b in zip(o["our_loss"], o["ref_loss"]):
            assert abs(a - b) < 5e-5 * abs(b), (uf, o["our_loss"], o["ref_loss"])
        for a, b in zip(o["our_gnorm"], o["ref_gnorm"]):  # (the norm fairseq's own clip_grad_norm_ computed from p.grad)
            assert abs(a - b) < 2e-3 * b, (uf, o["our_gnorm"], o["ref_gnorm"])
        assert o["our_loss"][0] != o["our_loss"][-1]  # the parameters really moved between the forwards
        for la, lb in zip(o["our_log"], o["ref_log"]):
            for k, v in lb.items():
                assert abs(la[k] - v) < 5e-5 * max(1.0, abs(v)), (uf, k, la[k], v)
        # parameters after the updates: the bound of the package's own trajectory test against the oracle (tests/test_host.py)
        assert o["worst_param"][0] < 2e-4, (uf, o["worst_param"])
