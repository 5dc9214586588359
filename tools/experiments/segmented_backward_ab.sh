B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
# (S2ST_XSTREAM_VARIANT / S2ST_EXCHANGE_ON_SIDE / S2ST_SEGMENTED_ONLY were temporary switches of these experiments; removed after them)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3 4; do
  echo "== one backward call: $($B 2>/dev/null | line)"
  echo "== segment-wise backward calls, no hook work: $(S2ST_SEGMENTED_ONLY=1 $B 2>/dev/null | line)"
done
