#!/bin/bash
# round 5, VERDICT r4 item 1: the training step as two utterance-half chains (S2ST_CHAINS=2) against the one-chain schedule
OUT=gpurun_out/r05_chains_ab.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; }
for rep in 1 2; do
  for v in 1 2; do
    echo "== base_recipe S2ST_CHAINS=$v steps 100 (run $rep)" | tee -a $OUT
    S2ST_CHAINS=$v timeout 600 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line | tee -a $OUT
  done
done
for v in 1 2; do
  echo "== base_recipe S2ST_CHAINS=$v steps 20 warmup 3 (the driver's form)" | tee -a $OUT
  S2ST_CHAINS=$v timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line | tee -a $OUT
  echo "== base_recipe_hubert S2ST_CHAINS=$v steps 50" | tee -a $OUT
  S2ST_CHAINS=$v timeout 600 python bench.py --config base_recipe_hubert --steps 50 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line | tee -a $OUT
done
