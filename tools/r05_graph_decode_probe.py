"""Does a decode step replay as a hipGraph, and what does the host pay per step then?  torch.cuda.graph captures the C
library's launches of ONE step (step 5, its pointers baked in -- timing only, the replays recompute the same step); compared
with calling the step directly.   python tools/r05_graph_decode_probe.py [B]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import s2st_amd  # noqa
import s2st_oracle as O
from configs import CONFIGS
PKG = "speech-to-speech-translation_amd"
tasks = importlib.import_module(PKG + ".tasks")
G = importlib.import_module(PKG + ".speech_generator")
D = importlib.import_module(PKG + ".data")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
a = O.make_args(**CONFIGS["base_recipe"])
task = tasks.S2ST_TranslationTask.setup_task(a, device=dev)
model = task.build_model(a)
corpus = D.SyntheticFisherCorpus(n_utts=256, seed=9)
s = corpus.collate_batch(range(B))
s["net_input"]["collated_audios_orig"] = None
s["net_input"]["padding_mask"] = None
gen = G.AutoRegressiveSpeechGenerator(model, None, None, max_iter=40, eos_prob_threshold=2.0)
gen.generate(model, s)   # decode_begin + 40 steps: caches, buffers, workspace exist now
torch.cuda.synchronize()
eng = model.engine
N = 200


def direct():
    for _ in range(N):
        eng.decode_step_into(5, 12345, 2.0, 40)


for _ in range(2):
    direct()
torch.cuda.synchronize()
t0 = time.perf_counter(); direct(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("direct: host enqueue %.3f ms/step, until GPU done %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eng.decode_step_into(5, 12345, 2.0, 40)
    with torch.cuda.graph(g, stream=side):
        eng.decode_step_into(5, 12345, 2.0, 40)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("graph replay: host enqueue %.3f ms/step, until GPU done %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# two graphs replayed alternately on two streams (two decode chains)
s2 = torch.cuda.Stream()
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
    with torch.cuda.stream(s2):
        g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("the graph on two streams alternately: host %.3f ms per pair, GPU %.3f ms per pair" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
