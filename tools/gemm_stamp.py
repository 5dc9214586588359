"""Reads the per-workgroup clock stamps of a -DS2ST_GEMM_STAMP build (tools/gemm_stamp.sh): cycles from workgroup start to
(prologue issued, first stage landed + barrier, end of the K-loop, epilogue stores drained), averaged over the workgroups."""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(os.environ["S2ST_HIP_LIB"], emulator=False)
d = torch.device("cuda:0")
os.environ["S2ST_GEMM_TILE"] = "128x128"
g = torch.Generator().manual_seed(1)
for (M, N, K, akm, bkm, out) in ((4584, 2048, 512, True, True, "h"), (4584, 512, 2048, True, True, "h"), (4584, 512, 512, True, True, "h"),
                                 (4584, 512, 512, True, True, "f32+h"), (4584, 1536, 512, True, True, "h"), (4096, 4096, 4096, True, True, "h"),
                                 (4584, 512, 2048, True, False, "h"), (4584, 512, 2048, True, True, "bias+resid")):
    A = torch.randn(M, K, generator=g).bfloat16(); B = torch.randn(N, K, generator=g).bfloat16()
    Am = (A if akm else A.t().contiguous()).to(d); Bm = (B if bkm else B.t().contiguous()).to(d)
    Ch = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
    Cc = torch.zeros(M, N, device=d) if out != "h" else None
    kw = {}
    if out == "bias+resid":
        kw = dict(bias=torch.randn(N, generator=g).to(d), resid=torch.randn(M, N, generator=g).to(d))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ws = torch.zeros(tiles * 16 + 64, device=d)  # 8 int64 per workgroup
    for _ in range(3):
        bd.gemm(Am, Bm, Cc, M, N, K, a_kmajor=akm, b_kmajor=bkm, a_ld=Am.shape[1], b_ld=Bm.shape[1], c_bf16=Ch, ws=ws, **kw)
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[: tiles * 8].view(tiles, 8).cpu().double()
    t0 = st[:, 0]
    rel = [(st[:, i] - t0) for i in range(1, 6)]
    first = t0.min()
    span = (st[:, 4].max() - first)
    starts = (t0 - first)
    print("M %5d N %5d K %5d %s%s out %-5s tiles %4d | per workgroup (cycles, mean): prologue issued %5.0f, first stage landed %5.0f, "
          "K-loop done %6.0f (%5.0f per K-step), epilogue issued %6.0f, stores drained %6.0f" % (
              M, N, K, "K" if akm else "R", "K" if bkm else "R", out, tiles, rel[0].mean(), rel[1].mean(), rel[2].mean(),
              (rel[2] - rel[1]).mean() / ((K + 63) // 64), rel[3].mean(), rel[4].mean()), flush=True)
