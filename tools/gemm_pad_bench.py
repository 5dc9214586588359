#!/usr/bin/env python3
"""Does the row stride of the operands matter (L2 channel spread)?  GEMM timing with padded leading dimensions."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s2st_amd  # noqa
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library()
dev = torch.device("cuda:0")


def run(tag, M, N, K, akm, bkm, pad, iters=50):
    lda = (K if akm else M) + pad
    ldb = (K if bkm else N) + pad
    A = torch.randn((M if akm else K), lda, device=dev).to(torch.bfloat16)
    B = torch.randn((N if bkm else K), ldb, device=dev).to(torch.bfloat16)
    Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: bd.gemm(A, B, None, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=lda, b_ld=ldb, c_ld=N, c_bf16=Ch)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    print(f"{tag:10s} M{M:5d} N{N:5d} K{K:5d} {'K' if akm else 'R'}{'K' if bkm else 'R'} pad {pad:4d}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF/s", flush=True)


for pad in ([int(x) for x in os.environ.get('PADS', '0').split(',')]):
    run("fc1 fwd", 4584, 2048, 512, True, True, pad)
    run("fc2 fwd", 4584, 512, 2048, True, True, pad)
    run("sq4096", 4096, 4096, 4096, True, True, pad, iters=10)
    run("fc1 dgradR", 4584, 512, 2048, True, False, pad)
    run("wgrad RR", 2048, 512, 4584, False, False, pad)
    run("qkv fwd", 4584, 1536, 512, True, True, pad)
    run("out fwd", 4584, 512, 512, True, True, pad)
    run("dec fc1", 3120, 2048, 512, True, True, pad)
