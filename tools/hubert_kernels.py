"""Kernel table of one fast-mode HuBERT-base forward (24 x 8 s of audio) from the library's per-dispatch events, and its GPU
time by HIP events: python tools/hubert_kernels.py   (S2ST_HIP_LIB=<experimental build> S2ST_GEMM_TILE=256x128 to try a form)"""
import importlib, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import s2st_amd  # noqa
import hubert_oracle as HO
M = importlib.import_module("speech-to-speech-translation_amd.models.hubert")
bd = importlib.import_module("speech-to-speech-translation_amd.runtime.binding")
dev = torch.device("cuda:0")
f = M.HubertFrontend(dev, precise=False)
f.load_state_dict(HO.synth_state(HO.BASE))
wave, pad, _ = HO.synth_audio(24, 128000, 3)
wave = wave.to(dev)
for _ in range(5):
    f.extract_features(wave, pad)
torch.cuda.synchronize()
ts = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f.extract_features(wave, pad); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("GPU ms per call: median %.3f min %.3f" % (sorted(ts)[len(ts)//2], min(ts)))
lib = bd.lib()
lib.s2st_profile_enable.argtypes = [C.c_int32]
lib.s2st_profile_report.argtypes = [C.c_char_p, C.c_int64]
lib.s2st_profile_report.restype = C.c_int64
lib.s2st_profile_enable(1)
f.extract_features(wave, pad)
torch.cuda.synchronize()
lib.s2st_profile_enable(0)
buf = C.create_string_buffer(1 << 16)
lib.s2st_profile_report(buf, len(buf))
rows = []
for ln in buf.value.decode().splitlines():
    p = ln.split("\t")
    rows.append((float(p[2]), int(p[1]), p[0]))
for us, n, tag in sorted(rows, reverse=True)[:12]:
    print("%9.1f us %4d x %8.1f  %s" % (us, n, us / n, tag[:80]))
