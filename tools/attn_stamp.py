"""Reads the per-workgroup clock stamps of a -DS2ST_ATTN_STAMP build (tools/attn_stamp.sh): where a dK/dV workgroup of the
attention backward spends its time, on the encoder self-attention shape of the bench workload."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(os.environ["S2ST_HIP_LIB"], emulator=False)
d = torch.device("cuda:0")
for (B, H, T, S, causal) in ((16, 4, 213, 213, False), (16, 4, 283, 283, True), (16, 4, 283, 213, False)):
    dh, Cm = 128, 4 * 128
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, T, Cm, generator=g).bfloat16().to(d); k = torch.randn(B, S, Cm, generator=g).bfloat16().to(d)
    v = torch.randn(B, S, Cm, generator=g).bfloat16().to(d); dO = torch.randn(B, T, Cm, generator=g).to(d)
    nkx, nqx = (S + 63) // 64, (T + 63) // 64
    nblk = (nkx + nqx) * B * H
    scratch = torch.zeros(max(nblk * 16 + 64, B * H * T), device=d)
    for _ in range(3):
        scratch.zero_()
        bd.flash_attention(q, k, v, H, causal=causal, drop_p=0.1, seed=5, dO=dO, bf16_o=True, scratch=scratch, bf16_grads=True)
    torch.cuda.synchronize()
    st = scratch.view(torch.int64)[: nblk * 8].view(B * H, nkx + nqx, 8)[:, :nkx, :].reshape(-1, 8).cpu().double()
    st = st[st[:, 0] > 0]
    t0 = st[:, 0]
    rel = [(st[:, i] - t0).mean() for i in range(1, 6)]
    span = st[:, 5].max() - t0.min()
    ntiles = (T + 31) // 32
    print("B %d H %d T %d S %d causal %d: %d dK/dV workgroups; clock ticks (mean): first tile in LDS %.0f, second tile in LDS %.0f "
          "(one iteration: %.0f), of which scores+dP+exp+dropout %.0f; loop done %.0f (%d tiles), gradients stored %.0f; "
          "all workgroups span %.0f ticks" % (B, H, T, S, causal, st.shape[0], rel[0], rel[1], rel[1] - rel[0], rel[2] - rel[1],
                                              rel[3], ntiles, rel[4], span), flush=True)
