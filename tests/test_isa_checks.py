"""Build-time ISA checks of the hand-scheduled kernels (CPU container: hipcc cross-compiles gfx950 to assembly).

* csrc/s2st_asm.h issues `ds_read_b64_tr_b16` as inline assembly and orders it by hand; `tools/check_raw_reads.py` reads
  the generated ISA and fails when any instruction names a destination of such a read before the `s_waitcnt lgkmcnt(0)`
  that makes it valid (the check used to be run by hand).
* the CTC recursion's step must keep its emission prefetch in flight across the barrier: a counted `vmcnt` wait inside
  the loop, never `vmcnt(0)` in front of the raw `s_barrier`.
"""
import importlib.util
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech-to-speech-translation_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


SOURCES = ("gemm_bf16.hip", "gemm_bf16_w4.hip", "losses.hip")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    """The three sources compiled to gfx950 assembly, side by side (one hipcc process each)."""
    # (cached beside the build's objects, keyed by the sources' modification times: the three compilations take ~2 minutes
    #  of the CPU suite, the kernels change a few times a round)
    d = os.path.join(CSRC, "build", "isa")
    os.makedirs(d, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")]
    procs, res = {}, {}
    for src in SOURCES:
        out = os.path.join(d, src + ".s")
        newest = max(os.path.getmtime(p) for p in deps + [os.path.join(CSRC, src)])
        if os.path.exists(out) and os.path.getsize(out) > 0 and os.path.getmtime(out) > newest:
            res[src] = out
            continue
        procs[src] = (out, subprocess.Popen(
            [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-Wno-unused-value",
             "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", "-x", "hip", os.path.join(CSRC, src), "-o", out]))
    for src, (out, pr) in procs.items():
        if pr.wait() != 0:
            if os.path.exists(out):
                os.remove(out)
            raise AssertionError("hipcc failed on " + src)
        res[src] = out
    return res


pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="needs hipcc")


@pytest.mark.parametrize("src", ["gemm_bf16.hip", "gemm_bf16_w4.hip"])
def test_hand_issued_transposed_reads_are_waited_for(src, asm):
    spec = importlib.util.spec_from_file_location("check_raw_reads", os.path.join(ROOT, "tools", "check_raw_reads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, bad, msgs = mod.check(asm[src])
    assert total > 0, "no hand-issued transposed read found: the check looks at the wrong thing"
    assert bad == 0, "\n".join(msgs)


def test_ctc_step_keeps_its_prefetch_in_flight(asm):
    s = open(asm["losses.hip"]).read()
    m = re.search(r"^(_ZN\S*ctc_ab_kernelILi2E\S*):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M)
    assert m, "ctc_ab_kernel<2> not found"
    body = m.group(2).splitlines()
    # the recursion loop = the blocks between the loop header and the back edge that contain v_exp; inside them every
    # vmcnt wait must be a counted one (> 0)
    loop = [i for i, l in enumerate(body) if "Loop Header" in l]
    assert loop, "no loop found"
    start = loop[-1]
    end = max(i for i, l in enumerate(body) if "s_cbranch" in l and i > start)
    waits = [l.strip() for l in body[start:end] if l.strip().startswith("s_waitcnt") and "vmcnt" in l]
    assert waits, "no vmcnt wait in the loop (the prefetch is not consumed?)"
    assert all("vmcnt(0)" not in w for w in waits), waits
    assert sum("s_barrier" in l for l in body[start:end]) >= 1
