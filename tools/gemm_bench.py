#!/usr/bin/env python3
"""GEMM shape sweep on the GPU (kernel tuning aid): python tools/gemm_bench.py [f32|bf16|both]
Shapes are the ones one training step of the bench workload launches (gpurun_out/gemm_shapes.csv)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import s2st_amd  # noqa
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library()
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"


def run(tag, M, N, K, akm, bkm, accumulate=False, iters=30, batch=1, dt=torch.float32):
    r8 = lambda x: (x + 7) // 8 * 8
    A = torch.randn((batch, M, r8(K)) if akm else (batch, K, r8(M)), device=dev).to(dt)
    B = torch.randn((batch, N, r8(K)) if bkm else (batch, K, r8(N)), device=dev).to(dt)
    C = torch.zeros(batch, M, r8(N), device=dev)
    kw = dict(a_kmajor=akm, b_kmajor=bkm, accumulate=accumulate, a_ld=A.shape[2], b_ld=B.shape[2], c_ld=r8(N),
              batch=batch, a_zo=A.shape[1] * A.shape[2], b_zo=B.shape[1] * B.shape[2], c_zo=M * r8(N))
    if accumulate and dt == torch.bfloat16:
        kw["ws"] = WS
    f = lambda: bd.gemm(A, B, C, M, N, K, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    print(f"{tag:12s} {str(dt)[6:]:9s} M{M:6d} N{N:5d} K{K:6d} b{batch:3d} {'K' if akm else 'R'}{'K' if bkm else 'R'} acc{int(accumulate)}: "
          f"{t*1e6:8.1f} us  {2.0*M*N*K*batch/t/1e12:7.1f} TF/s", flush=True)


WS = torch.empty(16 << 20, device=dev)
Mr, Md = 4584, 3120
cases = [
    ("fc1 fwd", Mr, 2048, 512, True, True, False, 1), ("fc2 fwd", Mr, 512, 2048, True, True, False, 1),
    ("qkv fwd", Mr, 1536, 512, True, True, False, 1), ("out fwd", Mr, 512, 512, True, True, False, 1),
    ("fc1 dgrad", Mr, 512, 2048, True, False, False, 1), ("fc2 dgrad", Mr, 2048, 512, True, False, False, 1),
    ("qkv dgrad", Mr, 512, 1536, True, False, False, 1), ("out dgrad", Mr, 512, 512, True, False, False, 1),
    ("fc1 wgrad", 2048, 512, Mr, False, False, True, 1), ("fc2 wgrad", 512, 2048, Mr, False, False, True, 1),
    ("qkv wgrad", 1536, 512, Mr, False, False, True, 1), ("out wgrad", 512, 512, Mr, False, False, True, 1),
    ("dec out", Md, 512, 512, True, True, False, 1), ("post conv", Md, 512, 2560, True, True, False, 1),
    ("post wgrad", 512, 2560, Md, False, False, True, 1),
    ("QK^T", 191, 191, 128, True, True, False, 96), ("PV", 191, 128, 191, True, False, False, 96),
    ("dV", 191, 128, 191, False, False, False, 96),
    ("sq 4096", 4096, 4096, 4096, True, True, False, 1), ("sq 4096 RR", 4096, 4096, 4096, False, False, False, 1),
]
for dt in ([torch.float32] if which == "f32" else [torch.bfloat16] if which == "bf16" else [torch.float32, torch.bfloat16]):
    for c in cases:
        run(c[0], c[1], c[2], c[3], c[4], c[5], c[6], batch=c[7], dt=dt)
