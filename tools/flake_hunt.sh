# how often the two-rank bf16 test fails under each switch: bash tools/flake_hunt.sh <runs> "<VAR=val>" ...
n=$1; shift
for v in "" "$@"; do
  f=0
  for i in $(seq $n); do
    env $v timeout 120 python -m pytest tests/test_distributed.py -q -m gpu -k "bf16_mode" 2>&1 | grep -q "1 passed" || f=$((f+1))
  done
  echo "== [$v] failures $f / $n"
done
