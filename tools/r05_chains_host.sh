#!/bin/bash
OUT=gpurun_out/r05_chains_host.txt
: > $OUT
for v in 1 2; do
  echo "== S2ST_CHAINS=$v" >> $OUT
  S2ST_BENCH_VERBOSE=1 S2ST_CHAINS=$v timeout 600 python bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | grep -E "single step|GPU time on|timed region" >> $OUT
done
