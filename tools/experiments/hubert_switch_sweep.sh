# config 3 (frozen HuBERT front end beside the step): the GEMM-form switches on the final schedule, alternating
B="python bench.py --config base_recipe_hubert --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== default: $($B 2>/dev/null | line)"
  for v in 0 1; do echo "== S2ST_GEMM_P4=$v: $(S2ST_GEMM_P4=$v $B 2>/dev/null | line)"; done
  for v in 8 24; do echo "== S2ST_ADAM_CHUNKS=$v: $(S2ST_ADAM_CHUNKS=$v $B 2>/dev/null | line)"; done
done
