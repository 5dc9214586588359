# Config 5 (AR decode + Griffin-Lim, 64 utterances) with the key / value caches loaded nontemporally by the decode attention
# (S2ST_DECODE_KV_STREAM was a switch of the experiment; removed after it: profiles/r06_nontemporal_other_streams_ab.txt)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('utt/s', d['value'], 'early-stop', d.get('early_stop',{}).get('utterances_per_s'))"; }
for rep in 1 2 3 4; do
  for s in 0 1; do
    echo "== kv_stream $s: $(S2ST_DECODE_KV_STREAM=$s python bench.py --config infer_base --steps 8 --warmup 2 --cpu-seconds 0 --no-roofline 2>/dev/null | line)"
  done
done
