#!/bin/bash
# what bounds config 5: the same run with a cheaper vocoder (8 Griffin-Lim iterations instead of 64) by decode chains
OUT=gpurun_out/r05_infer_bound.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], 'utt/s; chains', c['decode_chains'], '; no overlap', c.get('value_without_vocoder_overlap'), '; batch0 decode ms', c.get('batch0_decode_ms'), '; vocoder alone ms', c.get('batch0_vocoder_alone_ms'))"; }
for it in 64 8; do
for ch in 1 2 3; do
  echo "== GL iterations $it, chains $ch" >> $OUT
  S2ST_BENCH_GL_ITERS=$it S2ST_DECODE_CHAINS=$ch timeout 600 python bench.py --config infer_base --no-other-configs 2>&1 | tail -1 | line >> $OUT 2>&1
done
done
