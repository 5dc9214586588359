"""Fused attention kernels at the bench's batch geometries (max-tokens 20000 batches: utterances x encoder / decoder lengths),
kernel time per launch from the library's per-dispatch events: how the short batches price against the long ones.
python tools/attn_shapes_bench.py"""
import importlib, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64
H, dh = 4, 128


def run(B, T, S, causal, reps=6):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, T, H * dh, generator=g).bfloat16().to(d)
    k = torch.randn(B, S, H * dh, generator=g).bfloat16().to(d)
    v = torch.randn(B, S, H * dh, generator=g).bfloat16().to(d)
    dO = torch.randn(B, T, H * dh, generator=g).to(d)
    klen = torch.full((B,), S, dtype=torch.int32, device=d)
    for _ in range(2):
        bd.flash_attention(q, k, v, H, klen=klen, causal=causal, drop_p=0.1, seed=3, dO=dO, bf16_grads=True, bf16_o=True)
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(1)
    for _ in range(reps):
        bd.flash_attention(q, k, v, H, klen=klen, causal=causal, drop_p=0.1, seed=3, dO=dO, bf16_grads=True, bf16_o=True)
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    n = _lib.s2st_profile_report(buf, len(buf))
    out = {}
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        f = line.split("\t")
        out[f[0].split("<")[0]] = float(f[2]) / int(f[1])
    return out


print("B x T x S (causal)            fwd us   bwd us    fwd+bwd GFLOP   workgroups bwd")
for (B, E, D) in ((16, 213, 140), (24, 191, 131), (32, 134, 92), (40, 108, 73), (64, 71, 48), (80, 59, 40), (184, 27, 19)):
    for (T, S, c, label) in ((E, E, False, "enc self"), (D, D, True, "dec self"), (D, E, False, "cross")):
        r = run(B, T, S, c)
        fl = 14.0 * B * H * T * S * dh * (0.5 if c else 1.0) / 1e9
        wg = B * H * ((S + 63) // 64 + (T + 63) // 64)
        bw = r.get("flash_bwd_short_kernel", r.get("flash_bwd_kernel", 0))
        form = "short" if "flash_bwd_short_kernel" in r else "two-pass"
        fw = r.get("flash_fwd_short_kernel", r.get("flash_fwd_kernel", 0))
        print("%3d x %3d x %3d %-9s %8.1f %8.1f %14.2f %10d  %s" % (B, T, S, label, fw, bw, fl, wg, form))
