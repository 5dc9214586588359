#!/bin/bash
# config 5: the default line (3 chains; vocoder + upload share the fourth hardware queue) three times, then 2 / 4 chains, then
# the pool's probe off
OUT=gpurun_out/r05_infer_chains3.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], 'utt/s; chains', c['decode_chains'], '; early stop', (c.get('early_stop') or {}).get('value'), '; device rng', c.get('value_with_device_phase_rng'), '; collisions', c.get('stream_collisions'))"; }
for i in 1 2 3; do
  echo "== default" >> $OUT
  timeout 600 python bench.py --config infer_base --no-other-configs 2>&1 | tail -1 | line >> $OUT 2>&1
done
for ch in 2 4; do
  echo "== chains $ch" >> $OUT
  S2ST_DECODE_CHAINS=$ch timeout 600 python bench.py --config infer_base --no-other-configs 2>&1 | tail -1 | line >> $OUT 2>&1
done
for ch in 2 3; do
echo "== probe off, chains $ch" >> $OUT
S2ST_STREAM_PROBE=0 S2ST_DECODE_CHAINS=$ch timeout 600 python bench.py --config infer_base --no-other-configs 2>&1 | tail -1 | line >> $OUT 2>&1
done
timeout 900 python -m pytest tests/test_inference.py tests/test_inference_mtl.py -q -m gpu 2>&1 | tail -2 >> $OUT
