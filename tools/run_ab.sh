# A/B of engine / GEMM switches on the bench workload: bash tools/run_ab.sh <out-file> "<VAR=val>" ...
out=$1; shift
: > $out
for v in "" "$@"; do
  echo "== $v" >> $out
  for r in 1 2; do
    env $v python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> $out
  done
done
cat $out
