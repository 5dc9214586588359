#!/usr/bin/env python3
"""GEMM shape sweep on the GPU (kernel tuning aid): python tools/gemm_bench.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s2st_amd  # noqa
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library()
dev = torch.device("cuda:0")

def run(tag, M, N, K, akm, bkm, accumulate=False, iters=20, **kw):
    A = torch.randn((M, K) if akm else (K, M), device=dev)
    B = torch.randn((N, K) if bkm else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    f = lambda: bd.gemm(A, B, C, M, N, K, a_kmajor=akm, b_kmajor=bkm, accumulate=accumulate, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    print(f"{tag:28s} M{M:6d} N{N:5d} K{K:6d} {'K' if akm else 'R'}{'K' if bkm else 'R'} acc{int(accumulate)}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s")

Mr = 4992
run("fc1 fwd", Mr, 2048, 512, True, True)
run("fc2 fwd", Mr, 512, 2048, True, True)
run("qkv fwd", Mr, 1536, 512, True, True)
run("out fwd", Mr, 512, 512, True, True)
run("fc1 dgrad", Mr, 512, 2048, True, False)
run("fc2 dgrad", Mr, 2048, 512, True, False)
run("qkv dgrad", Mr, 512, 1536, True, False)
run("fc1 wgrad", 2048, 512, Mr, False, False, True)
run("fc2 wgrad", 512, 2048, Mr, False, False, True)
run("qkv wgrad", 1536, 512, Mr, False, False, True)
run("out wgrad", 512, 512, Mr, False, False, True)
run("square 4096", 4096, 4096, 4096, True, True)
run("square 4096 RR", 4096, 4096, 4096, False, False)
