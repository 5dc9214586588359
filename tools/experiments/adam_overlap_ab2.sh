# second pass: more chunks, and config 3
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== in line: $($B 2>/dev/null | line)"
  for c in 16 32 64; do echo "== overlapped, $c chunks: $(S2ST_ADAM_OVERLAP=1 S2ST_ADAM_CHUNKS=$c $B 2>/dev/null | line)"; done
done
for rep in 1 2; do
  echo "== config 3, in line: $($B --config base_recipe_hubert 2>/dev/null | line)"
  for c in 16 64; do echo "== config 3, overlapped, $c chunks: $(S2ST_ADAM_OVERLAP=1 S2ST_ADAM_CHUNKS=$c $B --config base_recipe_hubert 2>/dev/null | line)"; done
done
