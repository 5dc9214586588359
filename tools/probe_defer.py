"""Where does the host block when the vocoder is deferred?  Wall-clock stamps around the stages of generate()."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
PKG = "speech-to-speech-translation_amd"
G = importlib.import_module(PKG + ".speech_generator")
V = importlib.import_module(PKG + ".vocoder")
D = importlib.import_module(PKG + ".data")
E = importlib.import_module(PKG + ".runtime.engine")
T0 = time.perf_counter()
def stamp(tag):
    print("%9.2f ms  %s" % ((time.perf_counter() - T0) * 1e3, tag), flush=True)
# wrap stages
orig_begin = E.Engine.decode_begin
def begin(self, *a, **k):
    stamp("decode_begin >")
    r = orig_begin(self, *a, **k)
    stamp("decode_begin <")
    return r
E.Engine.decode_begin = begin
orig_voc = G.SpeechGenerator._vocode
def voc(self, feats):
    stamp("vocode >")
    r = orig_voc(self, feats)
    stamp("vocode <")
    return r
G.SpeechGenerator._vocode = voc
orig_pf = V.GriffinLim.prefetch_phases
def pf(self, n):
    stamp("prefetch_phases >")
    r = orig_pf(self, n)
    stamp("prefetch_phases <")
    return r
V.GriffinLim.prefetch_phases = pf
orig_pe = E.Engine.postnet_eval
def pe(self, f):
    stamp("loop done, postnet >")
    return orig_pe(self, f)
E.Engine.postnet_eval = pe

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
import s2st_amd  # noqa
C_ = importlib.import_module(PKG + ".configs")
tasks = importlib.import_module(PKG + ".tasks")
a = C_.recipe_args("base_recipe")
task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
torch.manual_seed(1)
model = task.build_model(a)
voc = V.GriffinLimVocoder(spec_bwd_max_iter=64, device=dev, sample_rate=24000, win_size=1200, hop_size=300, n_fft=2048, n_mels=80,
                          f_min=20, f_max=8000)
corpus = D.SyntheticFisherCorpus(n_utts=64, seed=1234)
order = np.argsort(-corpus.src_n_frames, kind="stable")
s_ = corpus.collate_batch(order.tolist())
s_["net_input"]["collated_audios_orig"] = None
s_["net_input"]["padding_mask"] = None
samples = [s_]
gens = [G.AutoRegressiveSpeechGenerator(model, voc, None, max_iter=int(s_["target_lengths"].max()), eos_prob_threshold=2.0)]
held = None
for i in range(6):
    stamp("generate %d >" % i)
    fin = gens[0].generate(model, samples[0], defer_vocoder=True)
    stamp("generate %d <" % i)
    if held is not None:
        held.wait()
    held = fin
held.wait()
torch.cuda.synchronize()
stamp("all done")
