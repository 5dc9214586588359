python -m pytest tests/test_inference.py tests/test_inference_mtl.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r04_n_tests.txt
nproc >> gpurun_out/r04_n_tests.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r04_n_tests.txt
for how in host numpy device off; do echo "== S2ST_GL_PHASE_STREAM=$how"; S2ST_GL_PHASE_STREAM=$how python bench.py --config infer_base 2>gpurun_out/r04_n_$how.err | tail -1; done > gpurun_out/r04_n_infer.txt
python - <<'PY' >> gpurun_out/r04_n_tests.txt 2>&1
import importlib, sys, time, numpy as np
sys.path.insert(0,'.')
voc = importlib.import_module("speech-to-speech-translation_amd.vocoder")
import torch
np.random.seed(1)
n = 45_600_000
st=np.random.get_state()
w = np.zeros(625, dtype=np.uint32); w[:624]=st[1]; w[624]=st[2]
buf = torch.zeros(n, dtype=torch.float64).pin_memory()
for T in (1,2,4,8,16):
    b = np.zeros((T+1,625),dtype=np.uint32)
    t=time.time(); voc._mt_host(w, n, None, b, T); ds=time.time()-t
    t=time.time(); voc._mt_host(w, n, buf.data_ptr(), b, T); dt=time.time()-t
    print(T, "threads: skip-only", round(ds*1e3,1), "ms; full", round(dt*1e3,1), "ms")
rs=np.random.RandomState(1)
x = buf.numpy()
t=time.time(); x[:] = rs.random_sample(n); print("numpy", round((time.time()-t)*1e3,1), "ms")
PY
cat gpurun_out/r04_n_tests.txt; grep -o '"value": [0-9.]*\|== .*\|"batch0_[a-z_]*": [0-9.]*' gpurun_out/r04_n_infer.txt
