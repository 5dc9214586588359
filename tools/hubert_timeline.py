"""Every dispatch of one fast-mode HuBERT-base forward (24 x 8 s of audio) in launch order with its own GPU duration
(the library's per-dispatch events): python tools/hubert_timeline.py"""
import importlib, os, sys, ctypes as C
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
lib = bd.lib()
lib.s2st_profile_enable.argtypes = [C.c_int32]
lib.s2st_profile_timeline.argtypes = [C.c_char_p, C.c_int64]
lib.s2st_profile_timeline.restype = C.c_int64
lib.s2st_profile_enable(1)
f.extract_features(wave, pad)
torch.cuda.synchronize()
lib.s2st_profile_enable(0)
buf = C.create_string_buffer(1 << 18)
lib.s2st_profile_timeline(buf, len(buf))
tot = 0.0
for i, ln in enumerate(buf.value.decode().splitlines()):
    tag, st, t0, dur = ln.split("\t")
    tot += float(dur)
    print("%3d  %9.1f  %8.1f us  %s" % (i, float(t0), float(dur), tag[:90]))
print("sum of durations %.1f us" % tot)
