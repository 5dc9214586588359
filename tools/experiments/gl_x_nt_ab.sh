# Griffin-Lim (64 iterations, 64 utterances batched): the projected spectra (236 MB per iteration, written by gl_stft_project_kernel,
# (S2ST_GL_X_NT was a switch of the experiment; removed after it: profiles/r06_nontemporal_other_streams_ab.txt)
# read once by gl_istft_ola_kernel) stored nontemporally (switch S2ST_GL_X_NT of the experiment), alternating
for rep in 1 2 3; do
  for x in 0 1; do echo "== x_nt $x: $(S2ST_GL_X_NT=$x python tools/infer_bench.py 64 60 2>/dev/null | grep 'batched over')"; done
done
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('utt/s', d['value'])"; }
for rep in 1 2 3 4; do
  for x in 0 1; do echo "== config 5, x_nt $x: $(S2ST_GL_X_NT=$x python bench.py --config infer_base --steps 8 --warmup 2 --cpu-seconds 0 --no-roofline 2>/dev/null | line)"; done
done
