"""Bit-level checksums of the fused attention backward (bf16 gradient outputs, nano-like shapes) for cold-process
comparison: python tools/attn_determinism_probe.py  (run it several times; every line must repeat exactly)."""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
for (B, T, S, H, dh, causal) in ((4, 23, 23, 2, 64, False), (4, 19, 19, 2, 64, True), (4, 19, 23, 2, 64, False), (3, 70, 70, 2, 64, True), (8, 117, 117, 4, 128, False)):
    g = torch.Generator().manual_seed(B * T + S)
    Cm = H * dh
    q = torch.randn(B, T, Cm, generator=g).to(torch.bfloat16).to(d)
    k = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16).to(d)
    v = torch.randn(B, S, Cm, generator=g).to(torch.bfloat16).to(d)
    dO = torch.randn(B, T, Cm, generator=g).to(d)
    klen = torch.tensor([max(1, S - 3 * i) for i in range(B)], dtype=torch.int32).to(d)
    sums = []
    for rep in range(3):
        out = bd.flash_attention(q, k, v, H, klen=klen, causal=causal, dO=dO, bf16_grads=True)
        torch.cuda.synchronize()
        dqh, dkh, dvh = out[5][0], out[5][1], out[5][2]
        sums.append(tuple(int(x.view(torch.int16).to(torch.int64).sum()) for x in (dqh, dkh, dvh)) + (float(out[0].double().sum()),))
    print(B, T, S, H, dh, causal, sums[0], "same-in-process" if sums[0] == sums[1] == sums[2] else ("DIFF " + str(sums[1:])))
