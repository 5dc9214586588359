"""Isolated timing of one decoding step's attention launch (s2st_decode_attn_f32) over key counts, cache types and the
distribution of key lengths in the batch: python tools/decode_attn_bench.py   (kernel time from events attached to each
dispatch; 6 rotating caches -- the decoder's six layers -- so that a launch does not find its rows in L2)"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
lib = bd.lib()
lib.s2st_profile_enable.argtypes = [C.c_int32]
lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
lib.s2st_profile_report.restype = C.c_int64


def kernel_us(fn, reps=30):
    for _ in range(6):
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
        if "decode_attn" in tag:
            tot += float(us)
            n += int(cnt)
    return tot / max(n, 1)


B, H, dh, S = 64, 4, 128, 320
Cd = H * dh
q = torch.randn(B, Cd, device=d)
o = torch.empty(B, Cd, device=d)
print("B %d, H %d, head width %d, rows of [K | V] %d floats" % (B, H, dh, 2 * Cd))
for bf in (0, 1):
    caches = [torch.randn(B, S, 2 * Cd, device=d).to(torch.bfloat16 if bf else torch.float32) for _ in range(6)]
    for label, kl in (("all 1", np.full(B, 1)), ("all 64", np.full(B, 64)), ("all 128", np.full(B, 128)),
                      ("all 256", np.full(B, 256)), ("all 315", np.full(B, 315)),
                      ("Fisher-shaped (mean 89, max 315)", np.minimum(315, np.maximum(8, (np.random.RandomState(1).gamma(2.0, 45.0, B)).astype(int))))):
        kl = kl.astype(np.int32)
        kl[0] = max(kl[0], kl.max())
        kd = torch.from_numpy(kl).to(d)
        f = lambda i: bd.call("s2st_decode_attn_f32", q, Cd, caches[i % 6], caches[i % 6][0, 0, Cd:], 2 * Cd, S * 2 * Cd, kd, S, B, H, dh,
                              dh ** -0.5, o, Cd, None, 0, None, None, 0, 0, bf)
        mb = float(kl.sum()) * Cd * 2 * (2 if bf else 4) / 1e6
        us = kernel_us(f)
        print("%-5s key lengths %-34s %7.2f us   %6.1f MB  %6.0f GB/s" % ("bf16" if bf else "fp32", label, us, mb, mb / us * 1e-3 * 1e3))
