"""Is there a sawtooth over M (rounds of tiles over the CUs) in the step's data-path products?  For (N, K, epilogue) of the
four biggest populations: kernel time (events on the dispatch, s2st_profile) for M = 2560 .. 4736 in steps of 64 with the
launcher's own pick, and the tile it picked.  usage: python tools/gemm_m_sweep.py"""
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
MMAX = 4736


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


for (N, K, epi) in ((2048, 512, "h"), (512, 2048, "br"), (1536, 512, "h"), (512, 512, "br"), (1024, 512, "h")):
    g = torch.Generator().manual_seed(N + K)
    sets = []
    for _ in range(3):
        A = (torch.rand(MMAX, K, generator=g) * 2 - 1).bfloat16().to(d)
        B = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).bfloat16().to(d)
        Ch = torch.zeros(MMAX, N, dtype=torch.bfloat16, device=d)
        Cf = torch.zeros(MMAX, N, device=d)
        R = torch.randn(MMAX, N, generator=g).to(d)
        bias = torch.randn(N, generator=g).to(d)
        sets.append((A, B, Ch, Cf, R, bias))
    print(f"== N {N} K {K} epilogue {epi}")
    for M in range(2560, MMAX + 1, 64):
        def fn(i):
            A, B, Ch, Cf, R, bias = sets[i % 3]
            if epi == "h":
                bd.gemm(A, B, None, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K, c_bf16=Ch, bias=bias, act=1, drop_p=0.1, seed=5)
            else:
                bd.gemm(A, B, Cf, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K, bias=bias, resid=R)
        fn(0); fn(1)
        res = []
        tags = None
        for r in range(3):
            us, tags = kernel_us(fn)
            res.append(us)
        us = statistics.median(res)
        print(f"M {M:5d}  {us:6.2f} us  {us / M * 128:6.3f} us per 128 rows  {2.0 * M * N * K / us / 1e6:5.0f} TF  {tags[0][:60] if tags else '?'}", flush=True)
