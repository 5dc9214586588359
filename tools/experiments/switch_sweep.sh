# the existing tuning switches re-visited on the round's final schedule (overlapped nontemporal update), two alternations on one box
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2; do
  echo "== default: $($B 2>/dev/null | line)"
  for v in 3 4 8; do echo "== S2ST_WGRAD_GROUP=$v: $(S2ST_WGRAD_GROUP=$v $B 2>/dev/null | line)"; done
  for v in 0 1 2; do echo "== S2ST_GEMM_W4=$v: $(S2ST_GEMM_W4=$v $B 2>/dev/null | line)"; done
  for v in 64 256; do echo "== S2ST_SPLITK_TARGET=$v: $(S2ST_SPLITK_TARGET=$v $B 2>/dev/null | line)"; done
  for v in 0 2; do echo "== S2ST_ATTN_GFUSE=$v: $(S2ST_ATTN_GFUSE=$v $B 2>/dev/null | line)"; done
  echo "== S2ST_ATTN_SHORT=0: $(S2ST_ATTN_SHORT=0 $B 2>/dev/null | line)"
  echo "== S2ST_NO_KV_HOIST=1: $(S2ST_NO_KV_HOIST=1 $B 2>/dev/null | line)"
  echo "== S2ST_NO_AUX_OVERLAP=1: $(S2ST_NO_AUX_OVERLAP=1 $B 2>/dev/null | line)"
done
