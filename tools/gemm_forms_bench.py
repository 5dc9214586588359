"""Isolated timing of the bf16 GEMM forms on the shapes of a training step: the 8-wave ring kernels (one workgroup per
CU) against the 4-wave early-release form (gemm_bf16_w4.hip, 2 - 3 workgroups per CU), per (shape, layout, epilogue).
Every form runs on the same rotating operand sets (8 sets: the operands of a launch are not the ones the previous launch
left in L2), interleaved round by round in ONE process (cdna_hip_programming.md 5.4 rule 24); median and minimum over
the rounds.   usage: python tools/gemm_forms_bench.py [--rounds 7] [--reps 16]"""
import argparse, importlib, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")

FORMS = [  # name, environment
    ("auto", {}),
    ("r8 auto", {"S2ST_GEMM_W4": "0"}),
    ("r8 128x128", {"S2ST_GEMM_TILE": "128x128"}),
    ("r8 128x64", {"S2ST_GEMM_TILE": "128x64"}),
    ("r8 64x64", {"S2ST_GEMM_TILE": "64x64"}),
    ("w4 128x128", {"S2ST_GEMM_W4": "1", "S2ST_GEMM_TILE": "128x128"}),
    ("w4 128x64", {"S2ST_GEMM_W4": "1", "S2ST_GEMM_TILE": "128x64"}),
]
KEYS = sorted({k for _, e in FORMS for k in e})
NSETS = 8


def make_sets(M, N, K, akm, bkm, epi):
    g = torch.Generator().manual_seed(M + N + K)
    sets = []
    for _ in range(NSETS):
        A = torch.randn(M, K, generator=g).bfloat16(); B = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
        Am = (A if akm else A.t().contiguous()).to(d); Bm = (B if bkm else B.t().contiguous()).to(d)
        kw = dict(a_kmajor=akm, b_kmajor=bkm, a_ld=Am.shape[1], b_ld=Bm.shape[1])
        if epi == "h":
            C = None; kw["c_bf16"] = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
        elif epi == "f32":
            C = torch.zeros(M, N, device=d)
        else:  # bias + residual -> fp32 (out-proj / fc2 of a layer)
            C = torch.zeros(M, N, device=d)
            kw["bias"] = torch.randn(N, device=d); kw["resid"] = torch.randn(M, N, device=d)
        sets.append((Am, Bm, C, kw))
    return sets


import ctypes as C
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64


def kernel_us(sets, M, N, K, reps):
    """mean begin -> end time of the launches (events attached to each dispatch: the figure a kernel trace reports),
    not the host's launch rate"""
    _lib.s2st_profile_enable(1)
    run(sets, M, N, K, reps)
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    n = _lib.s2st_profile_report(buf, len(buf))
    tot, cnt = 0.0, 0
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        f = line.split("\t")
        cnt += int(f[1]); tot += float(f[2])
    return tot / max(cnt, 1)


def run(sets, M, N, K, reps):
    for i in range(reps):
        Am, Bm, C, kw = sets[i % NSETS]
        bd.gemm(Am, Bm, C, M, N, K, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=16)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    shapes = []
    for M in ((4584,) if a.quick else (4584, 3408, 2800)):
        for (N, K, epi) in ((1536, 512, "h"), (512, 512, "br"), (2048, 512, "h"), (512, 2048, "br"), (512, 2048, "h"),
                            (2048, 512, "f32"), (1024, 512, "h"), (512, 512, "h")):
            shapes.append((M, N, K, True, True, epi))
    shapes.append((9168, 512, 512, True, True, "br"))
    shapes.append((9168, 512, 2048, True, True, "br"))
    shapes.append((4584, 512, 2048, True, False, "h"))
    shapes.append((4584, 2048, 512, True, False, "h"))
    print("%-34s " % "shape (us: median / min)" + " ".join("%15s" % n for n, _ in FORMS) + "   best", flush=True)
    totals = [0.0] * len(FORMS)
    for (M, N, K, akm, bkm, epi) in shapes:
        sets = make_sets(M, N, K, akm, bkm, epi)
        ts = [[] for _ in FORMS]
        for rnd in range(a.rounds + 1):
            for fi, (_, env) in enumerate(FORMS):
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                run(sets, M, N, K, 4)
                torch.cuda.synchronize()
                t = kernel_us(sets, M, N, K, a.reps)
                if rnd:
                    ts[fi].append(t)
        med = [statistics.median(t) for t in ts]
        for i, m in enumerate(med):
            totals[i] += m
        print("M %5d N %5d K %5d %s%s %-4s  " % (M, N, K, "K" if akm else "R", "K" if bkm else "R", epi) +
              " ".join("%7.1f /%6.1f" % (m, min(t)) for m, t in zip(med, ts)) + "   " + FORMS[med.index(min(med))][0], flush=True)
        del sets
    print("%-34s " % "sum of medians" + " ".join("%15.1f" % t for t in totals))
    for k in KEYS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
