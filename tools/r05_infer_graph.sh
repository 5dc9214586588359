#!/bin/bash
# config 5 with the decode step as one HIP-graph launch (S2ST_DECODE_GRAPH=1, default) against the step-by-step calls
OUT=gpurun_out/r05_infer_graph.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], 'utt/s; chains', c['decode_chains'], '; device rng', c.get('value_with_device_phase_rng'), '; no overlap', c.get('value_without_vocoder_overlap'), '; batch0 decode ms', c.get('batch0_decode_ms'), '; collisions', c.get('stream_collisions'))"; }
for g in 1 0 direct 1 0; do
  echo "== S2ST_DECODE_GRAPH=$g" >> $OUT
  S2ST_DECODE_GRAPH=$g timeout 600 python bench.py --config infer_base --no-other-configs 2>&1 | tail -1 | line >> $OUT 2>&1
done
for ch in 1 2 4; do
  echo "== graph, chains $ch" >> $OUT
  S2ST_DECODE_CHAINS=$ch timeout 600 python bench.py --config infer_base --no-other-configs 2>&1 | tail -1 | line >> $OUT 2>&1
done
