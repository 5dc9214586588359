# the gradient exchange's stand-in on the engine's SECOND stream (behind the bucket's weight gradients) instead of a third stream of its own
# (S2ST_XSTREAM_VARIANT / S2ST_EXCHANGE_ON_SIDE / S2ST_SEGMENTED_ONLY were temporary switches of these experiments; removed after them)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('exchange_proxy') or {}; print('ms_per_step', d['ms_per_step'], 'exposed', e.get('exposed_ms_mean', e.get('exposed_ms')))"; }
for rep in 1 2 3; do
  echo "== no exchange: $($B 2>/dev/null | line)"
  for p in 64,1,600 64,8,600 64,8,300 32,8,300; do
    echo "== own stream, proxy $p: $($B --exchange-proxy $p 2>/dev/null | line)"
    echo "== second stream, proxy $p: $(S2ST_EXCHANGE_ON_SIDE=1 $B --exchange-proxy $p 2>/dev/null | line)"
  done
done
