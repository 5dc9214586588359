#!/bin/bash
# the attention backward's forms that exist as switches, in the step (bench.py --steps 100) and per launch
OUT=gpurun_out/r05_attn_bwd_ab.txt
: > $OUT
for sw in "X=1" "S2ST_ATTN_BWD_SPLIT=1" "S2ST_ATTN_NW=2" "S2ST_ATTN_DVEC_KERNEL=1" "S2ST_ATTN_XCD=0"; do
  for rep in 1 2; do
    env $sw python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sw', d['ms_per_step'], d['value'])" >> $OUT
  done
done
