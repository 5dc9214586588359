#!/usr/bin/env python3
"""Host-side cost per launch (tuning aid)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s2st_amd  # noqa
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library()
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6
print("s2st_dropout_f32 via ctypes: %.2f us/launch" % t(lambda: bd.call("s2st_dropout_f32", x, x, 1 << 16, 1.0, 0.0, 0, 0)))
A = torch.randn(256, 512, device=dev).to(torch.bfloat16); B = torch.randn(512, 512, device=dev).to(torch.bfloat16)
Cc = torch.zeros(256, 512, device=dev)
print("gemm bf16 via ctypes:      %.2f us/launch" % t(lambda: bd.gemm(A, B, Cc, 256, 512, 512)))
Af = A.float(); Bf = B.float()
print("gemm f32 via ctypes:       %.2f us/launch" % t(lambda: bd.gemm(Af, Bf, Cc, 256, 512, 512)))
print("torch add_:                %.2f us/launch" % t(lambda: x.add_(1.0)))
e = torch.cuda.Event()
s2 = torch.cuda.Stream()
print("event record+wait:         %.2f us/pair" % t(lambda: (e.record(), s2.wait_event(e))))
print("torch zero_ (memset):      %.2f us" % t(lambda: x.zero_()))
