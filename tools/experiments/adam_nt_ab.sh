# Adam / gradient-norm kernels: cache policy and walk order A/B (round 6, second session).
#   S2ST_ADAM_VARIANT: 0 plain, 1 nontemporal fp32 streams, 2 plain x2 in flight, 3 nontemporal x2, 4 nontemporal x4,
#                      5 nontemporal + arena walked from its end, 6 plain + walked from its end
for v in 0 1 5 6; do S2ST_ADAM_VARIANT=$v python tools/adam_bench.py; done
for rep in 1 2 3 4; do
for v in 0 1 5 6; do echo "== variant $v: $(S2ST_ADAM_VARIANT=$v S2ST_SUMSQ_NT=0 python bench.py --steps 20 --warmup 5 --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])")"; done
done
