#!/bin/bash
# round 5 probe: what a half-size batch costs (the lower bound of one chain of a two-chain schedule)
out=gpurun_out/r05_halfbatch.txt
: > $out
for mt in 20000 10000 5000; do
  for ss in 0 1; do
    echo "== max-tokens $mt S2ST_NO_SIDE_STREAM=$ss" >> $out
    S2ST_NO_SIDE_STREAM=$ss python bench.py --steps 50 --warmup 5 --max-tokens $mt --cpu-seconds 0 --no-host-fed --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'], d['config'].get('global_batch_mel_frames'))
" >> $out
  done
done
cat $out
