# the gradient norm formed range by range behind the backward (GradReducer.enable_norm) against one pass in front of the optimizer
# (GradReducer.enable_norm / S2ST_NORM_AHEAD were a prototype of the experiment; reverted after it)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3 4 5; do
  echo "== one pass: $(S2ST_NORM_AHEAD=0 $B 2>/dev/null | line)"
  echo "== per range: $(S2ST_NORM_AHEAD=1 $B 2>/dev/null | line)"
done
S2ST_NORM_AHEAD=1 S2ST_BENCH_VERBOSE=1 $B 2>&1 | grep -E "sumsq_kernel|Traceback|Error" | head -5
