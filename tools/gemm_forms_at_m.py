"""Which form wins for the step's product shapes at the M of the 40 x 108 / 36 x 127 batches (just above a round boundary)?
Forced forms (S2ST_GEMM_TILE / S2ST_GEMM_W4) against the launcher's pick; kernel time from events on the dispatch.
usage: python tools/gemm_forms_at_m.py"""
import importlib, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as C
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64
FORMS = [("auto", {}), ("8w 128x128", {"S2ST_GEMM_TILE": "128x128", "S2ST_GEMM_W4": "0"}), ("8w 128x64", {"S2ST_GEMM_TILE": "128x64", "S2ST_GEMM_W4": "0"}),
         ("4w 128x128", {"S2ST_GEMM_TILE": "128x128", "S2ST_GEMM_W4": "1"}), ("4w 128x64", {"S2ST_GEMM_TILE": "128x64", "S2ST_GEMM_W4": "1"}),
         ("8w 64x64", {"S2ST_GEMM_TILE": "64x64", "S2ST_GEMM_W4": "0"})]
KEYS = ["S2ST_GEMM_TILE", "S2ST_GEMM_W4"]


def kernel_us(fn, reps=6):
    _lib.s2st_profile_enable(1)
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    n = _lib.s2st_profile_report(buf, len(buf))
    tot, cnt, tags = 0.0, 0, []
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        f = line.split("\t")
        cnt += int(f[1]); tot += float(f[2]); tags.append(f[0])
    return tot / max(cnt, 1), tags


print("%-34s " % "M x N x K epilogue" + " ".join("%16s" % n for n, _ in FORMS))
for (N, K, epi) in ((2048, 512, "h"), (512, 2048, "br"), (1536, 512, "h"), (512, 512, "br"), (1024, 512, "h")):
    for M in (2920, 3408, 4160, 4320, 4584, 4760, 4968):
        g = torch.Generator().manual_seed(N + K + M)
        sets = []
        for _ in range(3):
            A = (torch.rand(M, K, generator=g) * 2 - 1).bfloat16().to(d)
            B = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).bfloat16().to(d)
            sets.append((A, B, torch.zeros(M, N, dtype=torch.bfloat16, device=d), torch.zeros(M, N, device=d),
                         torch.randn(M, N, generator=g).to(d), torch.randn(N, generator=g).to(d)))

        def fn(i):
            A, B, Ch, Cf, R, bias = sets[i % 3]
            if epi == "h":
                bd.gemm(A, B, None, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K, c_bf16=Ch, bias=bias, act=1, drop_p=0.1, seed=5)
            else:
                bd.gemm(A, B, Cf, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K, bias=bias, resid=R)
        res = {n: [] for n, _ in FORMS}
        for r in range(4):
            for n, e in FORMS:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(e)
                us, _ = kernel_us(fn)
                if r:
                    res[n].append(us)
        for k in KEYS:
            os.environ.pop(k, None)
        print("%-34s " % f"{M} x {N} x {K} {epi}" + " ".join("%13.2f us" % statistics.median(res[n]) for n, _ in FORMS), flush=True)
