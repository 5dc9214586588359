# third pass: the overlapped update on fewer workgroups (a slower neighbour for longer)
# (S2ST_ADAM_BLOCKS was a temporary cap of the experiment; removed after it)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== in line: $(S2ST_ADAM_OVERLAP=0 $B 2>/dev/null | line)"
  for b in 1024 512 256 128; do echo "== overlapped 16 chunks, $b blocks: $(S2ST_ADAM_BLOCKS=$b $B 2>/dev/null | line)"; done
  echo "== overlapped 8 chunks, 256 blocks: $(S2ST_ADAM_BLOCKS=256 S2ST_ADAM_CHUNKS=8 $B 2>/dev/null | line)"
done
