#!/bin/bash
OUT=gpurun_out/r05_w4_e64.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; }
for f in 1.0 0.9 0.8 0.7; do
  echo "== S2ST_W4_E64=$f" >> $OUT
  S2ST_W4_E64=$f python tools/hubert_kernels.py 2>&1 | grep -E "GPU ms|w4_kernel" >> $OUT
  S2ST_W4_E64=$f timeout 600 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line >> $OUT
  S2ST_W4_E64=$f timeout 600 python bench.py --config base_recipe_hubert --steps 50 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line >> $OUT
done
