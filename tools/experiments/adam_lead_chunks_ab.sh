# the overlapped update's first three chunks made short (1/4, 1/4, 1/2 of an even share) against 16 even chunks, alternating
# (S2ST_ADAM_LEAD was a temporary switch of the experiment; removed after it)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3 4 5; do
  echo "== even chunks: $(S2ST_ADAM_LEAD=0 $B 2>/dev/null | line)"
  echo "== short leading chunks: $(S2ST_ADAM_LEAD=1 $B 2>/dev/null | line)"
done
