#!/usr/bin/env python3
"""Per-shape GEMM times of one fast-mode HuBERT-base forward (24 x 8 s): python tools/hubert_gemm_shapes.py"""
import collections, ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["S2ST_GEMM_PROFILE_DUMP"] = "/tmp/hub_shapes.csv"
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
for _ in range(3):
    f.extract_features(wave, pad)
torch.cuda.synchronize()
if os.path.exists("/tmp/hub_shapes.csv"):
    os.remove("/tmp/hub_shapes.csv")
lib = bd.lib()
lib.s2st_profile_gemm(1)
f.extract_features(wave, pad)
torch.cuda.synchronize()
fl, ms, n = C.c_double(), C.c_double(), C.c_long()
lib.s2st_profile_gemm_read(C.byref(fl), C.byref(ms), C.byref(n))
lib.s2st_profile_gemm(0)
agg = collections.defaultdict(lambda: [0, 0.0])
for line in open("/tmp/hub_shapes.csv"):
    r = line.strip().split(",")
    k = tuple(r[:8])
    agg[k][0] += 1
    agg[k][1] += float(r[8])
print("GEMM launches %d, %.2f ms, %.1f TFLOP/s" % (n.value, ms.value, fl.value / ms.value / 1e9))
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    M_, N_, K_, b_ = (int(x) for x in k[:4])
    print("M %7d N %5d K %5d b %3d %s tile %s  x%3d  %8.1f us each  %6.1f TF/s" % (M_, N_, K_, b_, k[4], k[6], c, us / c, 2.0 * M_ * N_ * K_ * b_ / (us / c) / 1e6))
