"""Clock stamps of the short-sequence attention backward (a -DS2ST_ATTN_STAMP build, tools/attn_short_stamp.sh): where one
(batch, head) workgroup spends its life.  Engine-like outputs: bf16 gradients + bias partial sums."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(os.environ["S2ST_HIP_LIB"], emulator=False)
d = torch.device("cuda:0")
names = ["start", "images written", "barrier 1", "phase A done", "dQ stored", "barrier 2", "dQ products", "dK dV stored (end)"]
for (B, H, T, S, causal) in ((40, 4, 108, 108, False), (40, 4, 73, 73, True), (40, 4, 73, 108, False), (64, 4, 71, 71, False), (184, 4, 27, 27, False), (16, 4, 128, 128, False)):
    dh, Cm = 128, 4 * 128
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, T, Cm, generator=g).bfloat16().to(d); k = torch.randn(B, S, Cm, generator=g).bfloat16().to(d)
    v = torch.randn(B, S, Cm, generator=g).bfloat16().to(d); dO = torch.randn(B, T, Cm, generator=g).to(d)
    nblk = B * H
    scratch = torch.zeros(max(nblk * 16 + 64, B * H * T), device=d)
    for _ in range(3):
        scratch.zero_()
        bd.flash_attention(q, k, v, H, causal=causal, drop_p=float(os.environ.get("DROP_P", "0.1")), seed=5, dO=dO, bf16_o=True, scratch=scratch, bf16_grads=False)
    torch.cuda.synchronize()
    st = scratch.view(torch.int64)[: nblk * 8].view(nblk, 8).cpu().double()
    st = st[st[:, 0] > 0]
    t0 = st[:, 0]
    rel = [(st[:, i] - t0).mean() for i in range(8)]
    span = st[:, 7].max() - t0.min()
    print("B %d H %d T %d S %d causal %d: %d workgroups; ticks since start (mean over workgroups): " % (B, H, T, S, causal, st.shape[0]) +
          ", ".join("%s %.0f" % (n, r) for n, r in zip(names[1:], rel[1:])) + "; first start -> last end %.0f ticks; starts spread %.0f" %
          (span, t0.max() - t0.min()), flush=True)
