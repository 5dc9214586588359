"""Isolated timing of the layer-norm backward row kernel (per-dispatch events) for the rows-per-wave variants, against a
plain device copy of the same bytes.   usage: python tools/ln_bwd_bench.py   (round 3 compared 1 / 2 / 4 rows per wave through S2ST_LN_RPW: within noise, the switch is gone)"""
import ctypes as C, importlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
    bd.load_library(bd.DEFAULT_LIB, emulator=False)
    lib = bd.lib()
    lib.s2st_profile_enable.argtypes = [C.c_int32]
    lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
    lib.s2st_profile_report.restype = C.c_int64
    d = torch.device("cuda:0")
    for rows in (4584, 3408):
        cols = 512
        sets = []
        for i in range(6):
            g = torch.Generator().manual_seed(i)
            x = torch.randn(rows, cols, generator=g).to(d); dy = torch.randn(rows, cols, generator=g).to(d)
            sets.append((x, dy, x.mean(1), 1.0 / (x.var(1, unbiased=False) + 1e-5).sqrt(), torch.zeros(rows, cols, device=d)))
        gamma = torch.ones(cols, device=d); dg = torch.zeros(cols, device=d); db = torch.zeros(cols, device=d)
        ns = int(bd._bind("s2st_layernorm_bwd_scratch")(rows, cols))
        scratch = torch.zeros(ns, device=d)
        def run(n):
            for i in range(n):
                x, dy, mu, rs, dx = sets[i % 6]
                bd.call("s2st_layernorm_bwd_f32", dy, x, gamma, mu, rs, dx, 1, dg, db, scratch, rows, cols)
        run(12); torch.cuda.synchronize()
        lib.s2st_profile_enable(1); run(60); torch.cuda.synchronize(); lib.s2st_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        n = lib.s2st_profile_report(buf, len(buf))
        out = []
        for line in buf.raw[:max(n, 0)].decode().splitlines():
            f = line.split("\t")
            out.append("%s %.2f us" % (f[0][:40], float(f[2]) / max(int(f[1]), 1)))
        # reference: a device copy moving the same 16 B per element (read 2 arrays + read-modify-write one)
        a = torch.empty(rows * cols * 3, device=d); b = torch.empty(rows * cols, device=d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            b.add_(a[: rows * cols]).add_(a[rows * cols: 2 * rows * cols])
        e1.record(); torch.cuda.synchronize()
        print("rows %d RPW %s: %s | two torch add_ passes %.2f us" % (rows, "by width", "; ".join(out), e0.elapsed_time(e1) * 1000 / 50))
    sys.exit(0)
subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ))  # (rows per wave: fixed by width since round 6)
