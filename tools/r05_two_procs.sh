#!/bin/bash
# round 5 probe: two PROCESSES stepping half-size batches at once on the one GPU -- an upper bound of what a
# two-chain schedule of the full batch could cost (each process also runs its own optimizer update)
out=gpurun_out/r05_two_procs.txt
mkdir -p gpurun_out
: > $out
run() { python bench.py --steps $2 --warmup 10 --max-tokens $1 --cpu-seconds 0 --no-host-fed --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$3', d['ms_per_step'], d['value'], d['config'].get('global_batch_mel_frames'))
"; }
echo "== solo 20000" >> $out; run 20000 100 solo >> $out
echo "== solo 10000" >> $out; run 10000 100 solo >> $out
for rep in 1 2; do
echo "== two at once, 10000 each (400 steps so that start-up skew is small)" >> $out
run 10000 400 procA >> $out & pa=$!
run 10000 400 procB >> $out & pb=$!
wait $pa $pb
done
echo "== three at once, 7000 each" >> $out
run 7000 400 procA >> $out & pa=$!
run 7000 400 procB >> $out & pb=$!
run 7000 400 procC >> $out & pc=$!
wait $pa $pb $pc
cat $out
