# What does the gradient-exchange stream cost the step BEFORE it moves a byte?  bench.py --exchange-proxy WGS,RANKS,GBPS with RANKS = 1 moves
# 2 (N - 1) / N = 0 bytes: no stand-in kernel is launched, only the stream's waits for both engine streams per bucket and the compute stream's
# wait for it after the backward.  Against the 8-rank stand-in (64 workgroups, 600 GB/s), with the runtime's default number of hardware queues and with 8.
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], (d.get('exchange_proxy') or {}).get('exposed_ms_mean'))"; }
for rep in 1 2 3; do
  echo "== no exchange stream: $($B 2>/dev/null | line)"
  echo "== waits only (RANKS 1): $($B --exchange-proxy 64,1,600 2>/dev/null | line)"
  echo "== 8-rank stand-in: $($B --exchange-proxy 64,8,600 2>/dev/null | line)"
  echo "== GPU_MAX_HW_QUEUES=8, no exchange stream: $(GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | line)"
  echo "== GPU_MAX_HW_QUEUES=8, waits only: $(GPU_MAX_HW_QUEUES=8 $B --exchange-proxy 64,1,600 2>/dev/null | line)"
  echo "== GPU_MAX_HW_QUEUES=8, 8-rank stand-in: $(GPU_MAX_HW_QUEUES=8 $B --exchange-proxy 64,8,600 2>/dev/null | line)"
done
