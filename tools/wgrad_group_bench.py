"""The grouped weight-gradient launch of one encoder / decoder layer (dW_i += dY_i^T X_i, K = tokens unsplit) ALONE, per
operand layout: "RR" = both operands [tokens][features] as the step has them (K is the slow index: transposed LDS reads,
256-byte row segments per DMA piece) against "KK" = K-contiguous copies [features][tokens] (what producing epilogues could
write).  Rotating operand sets (a launch does not find the previous launch's operands in L2), per-dispatch events.
python tools/wgrad_group_bench.py"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(os.environ.get("S2ST_HIP_LIB", bd.DEFAULT_LIB), emulator=False)
d = torch.device("cuda:0")
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64
NSETS = 6
ENC = [(1536, 512), (512, 512), (2048, 512), (512, 2048)]                                   # (M = dY width, N = X width)
DEC = [(1536, 512), (512, 512), (512, 512), (1024, 512), (2048, 512), (512, 2048)]          # + cross q, cross k|v


def sets_for(layer, K, kk):
    if kk == "panel":
        return sets_panel(layer, K)
    g = torch.Generator().manual_seed(K)
    out = []
    for _ in range(NSETS):
        probs, keep = [], []
        for (M, N) in layer:
            dY = (torch.randn(K, M, generator=g) * 0.1).bfloat16()
            X = torch.randn(K, N, generator=g).bfloat16()
            if kk:
                A, B = dY.t().contiguous().to(d), X.t().contiguous().to(d)
            else:
                A, B = dY.to(d), X.to(d)
            Cc = torch.zeros(M, N, device=d)
            keep += [A, B, Cc]
            probs.append(bd.gemm_args_bf16(A, B, Cc, M, N, K, a_kmajor=kk, b_kmajor=kk, a_ld=A.shape[1], b_ld=B.shape[1],
                                           accumulate=True))
        out.append((probs, keep))
    return out


def sets_panel(layer, K):
    """RR operands stored panel-major: [features / 128][K][128] (ld 128, panel stride K * 128)."""
    g = torch.Generator().manual_seed(K)
    out = []
    for _ in range(NSETS):
        probs, keep = [], []
        for (M, N) in layer:
            dY = (torch.randn(K, M, generator=g) * 0.1).bfloat16()
            X = torch.randn(K, N, generator=g).bfloat16()
            A = dY.view(K, M // 128, 128).permute(1, 0, 2).contiguous().to(d)
            B = X.view(K, N // 128, 128).permute(1, 0, 2).contiguous().to(d)
            Cc = torch.zeros(M, N, device=d)
            keep += [A, B, Cc]
            ga = bd.gemm_args_bf16(A, B, Cc, M, N, K, a_kmajor=False, b_kmajor=False, a_ld=128, b_ld=128, accumulate=True)
            ga.A.sp.bs = K * 128
            ga.B.sp.bs = K * 128
            probs.append(ga)
        out.append((probs, keep))
    return out


def run(layer, K, kk, reps=18):
    ss = sets_for(layer, K, kk)
    for i in range(NSETS):
        bd.gemm_group(ss[i][0])
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(1)
    for i in range(reps):
        bd.gemm_group(ss[i % NSETS][0])
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    n = _lib.s2st_profile_report(buf, len(buf))
    res = []
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        f = line.split("\t")
        res.append((f[0], float(f[2]) / int(f[1]), int(f[1])))
    return res


for name, layer in (("encoder layer", ENC), ("decoder layer", DEC)):
    for K in (2200, 3400, 4584):
        fl = sum(2.0 * M * N * K for M, N in layer)
        by = sum((M + N) * K * 2 + M * N * 8 for M, N in layer)
        tiles = sum(((M + 127) // 128) * ((N + 127) // 128) for M, N in layer)
        for kk in (False, True):  # ("panel": needs the panel-major hook of profiles/r06_wgrad_panel_major_probe.txt, not in the product)
            r = run(layer, K, kk)
            print("%s K %5d %s: %d tiles, %.1f GFLOP, %.0f MB  " % (name, K, "RR panel-major" if kk == "panel" else ("KK" if kk else "RR"), tiles, fl / 1e9, by / 1e6) +
                  "; ".join("%s %.1f us x %d (%.0f TFLOP/s, %.2f TB/s, %.0f cycles per K-step at 2.4 GHz)" %
                            (t, us, n, fl / us / 1e6, by / us / 1e6, us * 2400 / ((K + 63) // 64)) for t, us, n in r), flush=True)
