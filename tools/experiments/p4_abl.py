"""DEV: ablations of the 256 x 256 GEMM form (S2ST_P4_ABL: 1 no DMA in the loop, 2 no barriers, 3 no LDS reads, 4 no MFMAs)."""
import importlib, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as C
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64
os.environ["S2ST_GEMM_TILE"] = "256x256"

def run(M, N, K, reps=8):
    g = torch.Generator().manual_seed(1)
    sets = []
    for _ in range(4):
        A = (torch.rand(M, K, generator=g) * 2 - 1).bfloat16().to(d)
        B = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).bfloat16().to(d)
        sets.append((A, B, torch.zeros(M, N, dtype=torch.bfloat16, device=d)))
    out = {}
    for rnd in range(4):
        for abl in (0, 5, 6, 7):
            os.environ["S2ST_P4_ABL"] = str(abl)
            _lib.s2st_profile_enable(1)
            for i in range(reps):
                A, B, Ch = sets[i % 4]
                bd.gemm(A, B, None, M, N, K, a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K, c_bf16=Ch)
            torch.cuda.synchronize()
            _lib.s2st_profile_enable(0)
            buf = C.create_string_buffer(1 << 16)
            n = _lib.s2st_profile_report(buf, len(buf))
            tot = cnt = 0
            for line in buf.raw[:max(n, 0)].decode().splitlines():
                f = line.split("\t"); cnt += int(f[1]); tot += float(f[2])
            if rnd: out.setdefault(abl, []).append(tot / cnt)
    kt = K // 64
    print(f"{M}x{N}x{K}: " + "  ".join(f"abl{a} {statistics.median(v):7.1f} us ({statistics.median(v) / kt * 1000:5.0f} ns/K-tile)" for a, v in out.items()), flush=True)

for shp in ((4096, 4096, 4096), (8192, 8192, 4096)):
    run(*shp)
