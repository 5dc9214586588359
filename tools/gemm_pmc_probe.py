"""A few launches of the step's dominant products, for rocprofv3 --pmc passes (tools/gemm_pmc.sh)."""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
os.environ["S2ST_GEMM_TILE"] = "128x128"
g = torch.Generator().manual_seed(1)
for (M, N, K, akm, bkm) in ((4584, 2048, 512, True, True), (4584, 512, 2048, True, True), (4584, 512, 512, True, True),
                            (4096, 4096, 4096, True, True), (2048, 512, 4584, False, False)):
    A = torch.randn(M, K, generator=g).bfloat16(); B = torch.randn(N, K, generator=g).bfloat16()
    Am = (A if akm else A.t().contiguous()).to(d); Bm = (B if bkm else B.t().contiguous()).to(d)
    Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    for _ in range(10):
        bd.gemm(Am, Bm, None, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=Am.shape[1], b_ld=Bm.shape[1], c_bf16=Ch)
    torch.cuda.synchronize()
print("done")
