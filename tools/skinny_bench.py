"""Isolated timing of the AR-decoding products (s2st_gemm_skinny_f32 / s2st_ln_gemm_skinny_f32) per shape and batch size:
python tools/skinny_bench.py   (kernel time from events attached to each dispatch, 8 rotating weight sets so that the
weights of a launch are not the ones the previous launch left in L2)"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
lib = bd.lib()
lib.s2st_profile_enable.argtypes = [C.c_int32]
lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
lib.s2st_profile_report.restype = C.c_int64


def kernel_us(fn, reps=40):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    lib.s2st_profile_enable(1)
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    lib.s2st_profile_report(buf, len(buf))
    tot = n = 0
    for ln in buf.value.decode().splitlines():
        tag, cnt, us, _, _ = ln.split("\t")
        if "skinny" in tag:
            tot += float(us)
            n += int(cnt)
    return tot / max(n, 1)


SHAPES = [("qkv (LN)", 1536, 512, True), ("out-proj + resid", 512, 512, False), ("q (LN)", 512, 512, True),
          ("fc1 (LN, ReLU)", 2048, 512, True), ("fc2 + resid", 512, 2048, False), ("prenet 0", 256, 320, False),
          ("feat head (LN)", 320, 512, True), ("stop head (LN)", 1, 512, True)]
print("%-20s %6s %6s | %s" % ("product", "N", "K", "  ".join("M=%-3d us" % m for m in (16, 32, 64))))
for name, N, K, ln in SHAPES:
    row = []
    for M in (16, 32, 64):
        g = torch.Generator().manual_seed(N + K + M)
        ws = [(torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(d) for _ in range(8)]
        x = torch.randn(M, K, generator=g).to(d)
        y = torch.empty(M, N, device=d)
        b = torch.randn(N, generator=g).to(d)
        gam, bet = torch.ones(K, device=d), torch.zeros(K, device=d)
        r = torch.randn(M, N, generator=g).to(d)
        if ln:
            f = lambda i: bd.call("s2st_ln_gemm_skinny_f32", x, K, gam, bet, 1e-5, ws[i % 8], K, y, N, b, 0, M, N, K)
        else:
            f = lambda i: bd.call("s2st_gemm_skinny_f32", x, K, ws[i % 8], K, y, N, b, 0, 0.0, 0, r, N, M, N, K)
        row.append(kernel_us(f))
    print("%-20s %6d %6d | %s" % (name, N, K, "  ".join("%8.2f" % v for v in row)))
