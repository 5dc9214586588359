"""HuBERT's conv-stack products (k3 s2 512->512 over channel-last frames: rows = output frames, row pitch 2 * 512, K = 3 * 512,
GELU epilogue, bf16 result only) by GEMM form, in isolation: flat rows (no per-utterance split -- what a pitch-padded
layout gives) vs split rows (today's operand), W4 / ring / 256 x 256 four-phase forms.   python tools/conv_forms_bench.py"""
import importlib, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64

FORMS = [("auto", {}), ("P4 off", {"S2ST_GEMM_P4": "0"}), ("W4 128x128", {"S2ST_GEMM_P4": "0", "S2ST_GEMM_W4": "1", "S2ST_GEMM_TILE": "128x128"}),
         ("ring 128x128", {"S2ST_GEMM_P4": "0", "S2ST_GEMM_W4": "0"}), ("P4 forced", {"S2ST_GEMM_TILE": "256x256"})]
KEYS = sorted({k for _, e in FORMS for k in e})


def setenv(e):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(e)


def run(B, Tin, I, O, k, sd, flat, reps=6):
    Tout = (Tin - k) // sd + 1
    P_in = (Tin + 1) // 2 * 2
    g = torch.Generator().manual_seed(1)
    rows_in = B * (P_in if flat else Tin)
    A = (torch.rand(rows_in * I + k * I, generator=g) * 2 - 1).bfloat16().to(d)
    W = ((torch.rand(O, k * I, generator=g) * 2 - 1) / (k * I) ** 0.5).bfloat16().to(d)
    bias = torch.randn(O, device=d)
    M = B * (P_in // sd if flat else Tout)
    out = torch.zeros(M, O, dtype=torch.bfloat16, device=d)
    kw = dict(a_kmajor=True, b_kmajor=True, a_ld=sd * I, b_ld=k * I, bias=bias, act=2, c_bf16=out)
    if not flat:
        kw.update(a_per=Tout, a_bs=Tin * I)
    res = []
    for name, env in FORMS:
        setenv(env)
        tile = bd.gemm(A, W, None, M, O, k * I, return_tile=True, **kw)
        for _ in range(2):
            bd.gemm(A, W, None, M, O, k * I, **kw)
        torch.cuda.synchronize()
        _lib.s2st_profile_enable(1)
        for _ in range(reps):
            bd.gemm(A, W, None, M, O, k * I, **kw)
        torch.cuda.synchronize()
        _lib.s2st_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        n = _lib.s2st_profile_report(buf, len(buf))
        tot = cnt = 0
        tag = ""
        for line in buf.raw[:max(n, 0)].decode().splitlines():
            f = line.split("\t"); cnt += int(f[1]); tot += float(f[2]); tag = f[0]
        us = tot / max(cnt, 1)
        res.append("%-13s %7.1f us %6.0f TF/s  tile %s %s" % (name, us, 2.0 * M * O * k * I / us / 1e6, tile, tag.split("<")[0][-22:]))
    setenv({})
    return M, res


for (Tin, label) in ((25599, "conv1"), (12799, "conv2"), (6399, "conv3"), (3199, "conv4")):
    for flat in (False, True):
        M, res = run(24, Tin, 512, 512, 3, 2, flat)
        print("== %s  M %d  %s rows" % (label, M, "flat" if flat else "split"))
        for r in res:
            print("   " + r)
