# the overlapped update: the data path waits for the first k of the 16 chunks at once (k = 0: the shipped form; 16: in line), and other chunk counts
# (S2ST_ADAM_HOLD was a temporary switch of the experiment; removed after it)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  for k in 0 2 4 8; do echo "== hold $k: $(S2ST_ADAM_HOLD=$k $B 2>/dev/null | line)"; done
  for c in 10 12 20 24; do echo "== $c chunks: $(S2ST_ADAM_CHUNKS=$c $B 2>/dev/null | line)"; done
done
