#!/bin/bash
# round 5, VERDICT r4 item 5: convolution weight gradients over whole halo-image rows in the grouped launch
OUT=gpurun_out/r05_conv_wgrad.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; }
for rep in 1 2 3; do
  for v in 0 1 2; do
    echo "== S2ST_CONV_WGRAD_GROUP=$v (run $rep)" | tee -a $OUT
    S2ST_CONV_WGRAD_GROUP=$v timeout 600 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line | tee -a $OUT
  done
done
