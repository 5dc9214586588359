# third-update gradient norm of the micro model in FRESH processes, per switch: bash tools/cold_probe.sh <runs> "<VAR=val>" ...
n=$1; shift
for v in "" "$@"; do
  echo "== [$v]"
  for i in $(seq $n); do
    env REPS=1 NOSYNC=1 $v timeout 120 python tools/update_determinism_probe.py $PLAN 2>&1 | grep "rep  0" | sed 's/.*update: //' | awk '{print $3}' | tr '\n' ' '
  done
  echo
done
