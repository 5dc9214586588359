# The optimizer update overlapped with the next forward (round 3's S2ST_ADAM_OVERLAP, measured neutral then) re-measured with the
# nontemporal optimizer kernel, alternating on one box
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== in line: $($B 2>/dev/null | line)"
  for c in 4 8 16; do echo "== overlapped, $c chunks: $(S2ST_ADAM_OVERLAP=1 S2ST_ADAM_CHUNKS=$c $B 2>/dev/null | line)"; done
done
