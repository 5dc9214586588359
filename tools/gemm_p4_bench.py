"""The 256 x 256 four-phase bf16 GEMM (csrc/gemm_bf16_p4.hip) against the 128-row forms, in isolation: per shape the kernel
time (events attached to each dispatch) and TFLOP/s of (a) what the launcher picked before round 5 (S2ST_GEMM_P4=0),
(b) the new form forced (S2ST_GEMM_TILE=256x256), (c) the launcher's pick now.  Random operands (cdna_hip_programming.md
5.4 rule 25), 8 rotating operand sets, forms interleaved round by round in one process (rule 24); first a correctness check
of the forced form against an fp32 product on the host.   usage: python tools/gemm_p4_bench.py [--rounds 5] [--reps 8]"""
import argparse, importlib, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
bd.load_library(bd.DEFAULT_LIB, emulator=False)
d = torch.device("cuda:0")
import ctypes as C
_lib = bd.lib()
_lib.s2st_profile_enable.argtypes = [C.c_int32]
_lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
_lib.s2st_profile_report.restype = C.c_int64

FORMS = [("before (P4 off)", {"S2ST_GEMM_P4": "0"}), ("256x256 forced", {"S2ST_GEMM_TILE": "256x256"}), ("auto", {})]
KEYS = sorted({k for _, e in FORMS for k in e})
NSETS = 4


def setenv(e):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(e)


def make_sets(M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    sets = []
    for _ in range(NSETS):
        A = (torch.rand(M, K, generator=g) * 2 - 1).bfloat16().to(d)
        B = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).bfloat16().to(d)
        kw = dict(a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K)
        if epi == "h":
            Cc = None; kw["c_bf16"] = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
        else:
            Cc = torch.zeros(M, N, device=d)
            if epi == "br":
                kw["bias"] = torch.randn(N, device=d); kw["resid"] = torch.randn(M, N, device=d)
        sets.append((A, B, Cc, kw))
    return sets


def kernel_us(sets, M, N, K, reps):
    _lib.s2st_profile_enable(1)
    for i in range(reps):
        A, B, Cc, kw = sets[i % NSETS]
        bd.gemm(A, B, Cc, M, N, K, **kw)
    torch.cuda.synchronize()
    _lib.s2st_profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    n = _lib.s2st_profile_report(buf, len(buf))
    tot, cnt, tags = 0.0, 0, []
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        f = line.split("\t")
        cnt += int(f[1]); tot += float(f[2]); tags.append(f[0])
    return tot / max(cnt, 1), tags


def check():
    setenv({"S2ST_GEMM_TILE": "256x256"})
    worst = 0.0
    for (M, N, K, epi) in ((700, 520, 200, "f32"), (1024, 1024, 512, "br"), (1300, 1032, 264, "h"), (513, 257, 64, "f32")):
        g = torch.Generator().manual_seed(7)
        A = torch.randn(M, K, generator=g).bfloat16(); B = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
        ref = A.float() @ B.float().t()
        kw = dict(a_kmajor=True, b_kmajor=True, a_ld=K, b_ld=K)
        bias = res = None
        if epi == "br":
            bias = torch.randn(N, generator=g); res = torch.randn(M, N, generator=g)
            kw["bias"] = bias.to(d); kw["resid"] = res.to(d)
            ref = ref + bias + res
        if epi == "h":
            Cc = None; kw["c_bf16"] = torch.zeros(M, N, dtype=torch.bfloat16, device=d)
        else:
            Cc = torch.zeros(M, N, device=d)
        us, tags = kernel_us([(A.to(d), B.to(d), Cc, kw)] * NSETS, M, N, K, 1)
        out = (kw["c_bf16"].float() if epi == "h" else Cc).cpu()
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, err)
        print(f"check {M}x{N}x{K} {epi}: kernel {tags}, max rel err {err:.2e}")
        assert err < (1e-2 if epi == "h" else 2e-5), err
    print("correctness ok, worst", worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    check()
    shapes = [(4096, 4096, 4096, "h"), (8192, 8192, 4096, "h"),
              (4584, 2048, 512, "h"), (4584, 1536, 512, "h"), (3408, 2048, 512, "h"), (3408, 1536, 512, "h"),
              (4584, 2048, 512, "f32"), (9600, 3072, 768, "h"), (9600, 768, 3072, "br"), (9600, 2304, 768, "h"),
              (9600, 768, 768, "br"), (19200, 3072, 768, "h")]
    print("%-26s " % "shape" + " ".join("%28s" % n for n, _ in FORMS))
    for (M, N, K, epi) in shapes:
        sets = make_sets(M, N, K, epi)
        res = {n: [] for n, _ in FORMS}
        tags = {}
        for r in range(a.rounds + 1):
            for n, e in FORMS:
                setenv(e)
                us, tg = kernel_us(sets, M, N, K, a.reps)
                if r > 0:
                    res[n].append(us)
                tags[n] = tg
        fl = 2.0 * M * N * K
        print("%-26s " % f"{M}x{N}x{K} {epi}" + " ".join(
            "%9.1f us %7.0f TF %-7s" % (statistics.median(res[n]), fl / statistics.median(res[n]) / 1e6,
                                        (tags[n][0].split("<")[1].split(",")[0] + "x" + tags[n][0].split(",")[1].strip(" >"))[:7] if tags[n] else "?")
            for n, _ in FORMS), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
