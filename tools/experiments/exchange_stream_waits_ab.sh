B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
# (S2ST_XSTREAM_VARIANT / S2ST_EXCHANGE_ON_SIDE / S2ST_SEGMENTED_ONLY were temporary switches of these experiments; removed after them)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== no exchange stream: $($B 2>/dev/null | line)"
  for v in 0 1 2 3; do echo "== waits only, variant $v (1 = side wait, 2 = main event, 3 = both, 0 = none): $(S2ST_XSTREAM_VARIANT=$v $B --exchange-proxy 64,1,600 2>/dev/null | line)"; done
done
