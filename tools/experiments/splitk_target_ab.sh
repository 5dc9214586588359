# S2ST_SPLITK_TARGET (workgroups a split-K weight-gradient product is cut into: tiles x K ranges; 128 since round 2), alternating
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  for v in 128 192 256 384 512; do echo "== target $v: $(S2ST_SPLITK_TARGET=$v $B 2>/dev/null | line)"; done
done
for v in 128 256; do echo "== config 3, target $v: $(S2ST_SPLITK_TARGET=$v $B --config base_recipe_hubert 2>/dev/null | line)"; done
for v in 128 256; do echo "== config 3, target $v: $(S2ST_SPLITK_TARGET=$v $B --config base_recipe_hubert 2>/dev/null | line)"; done
