"""Isolated timing of the bf16 ring GEMM per (shape, layout, tile): input for the launcher's tile choice.
usage: python tools/gemm_tile_sweep.py"""
import os, sys, importlib, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")


def timeit(M, N, K, akm, bkm, tile, reps=30):
    os.environ["S2ST_GEMM_TILE"] = tile
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).bfloat16(); B = torch.randn(N, K, generator=g).bfloat16()
    Am = (A if akm else A.t().contiguous()).to(d); Bm = (B if bkm else B.t().contiguous()).to(d)
    Cc = torch.zeros(M, N, device=d); Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    kw = dict(a_kmajor=akm, b_kmajor=bkm, a_ld=Am.shape[1], b_ld=Bm.shape[1], c_bf16=Ch)
    for _ in range(3): bd.gemm(Am, Bm, Cc, M, N, K, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): bd.gemm(Am, Bm, Cc, M, N, K, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tiles = ["128x128", "128x64", "64x64", "256x128"]
print("%-28s " % "shape" + " ".join("%9s" % t for t in tiles) + "   best")
for M in (3120, 4584, 2400, 6000):
    for (N, K) in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (512, 1536), (1024, 512), (512, 1024)):
        for akm, bkm in ((True, True), (True, False)):
            ts = [timeit(M, N, K, akm, bkm, t) for t in tiles]
            print("M %5d N %5d K %5d %s%s  " % (M, N, K, "K" if akm else "R", "K" if bkm else "R") +
                  " ".join("%9.1f" % t for t in ts) + "   " + tiles[ts.index(min(ts))], flush=True)
