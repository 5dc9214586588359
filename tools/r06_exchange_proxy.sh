#!/bin/bash
# what a bandwidth-moving neighbour on a few CUs costs the training step -- the stand-in for the gradient
# all-reduce's kernels (bench.py --exchange-proxy WGS,RANKS,GBPS) against the plain step, alternating, on one box.
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 20 --warmup 3 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed "$@" 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read())
p = l.get('exchange_proxy')
print('%-18s ms/step %.3f' % ('$*'[-18:] if p else 'no proxy', l['ms_per_step']), ('exposed %.3f ms (max %.3f), %d buckets, %.0f MB moved per update' % (p['exposed_ms'], p['exposed_ms_max'], len(p['buckets_mib']), p['bytes_moved_per_update'] / 1e6)) if p else '')
"; }
for rep in 1 2; do
  run
  for w in 16 32 64; do for g in 150 300 600; do run --exchange-proxy $w,8,$g; done; done
done
run
