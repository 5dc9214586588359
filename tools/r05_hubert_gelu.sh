#!/bin/bash
OUT=gpurun_out/r05_hubert_gelu.txt
: > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; }
python tools/hubert_timeline.py 2>&1 | sed -n 2,12p >> $OUT
python tools/hubert_timeline.py 2>&1 | tail -1 >> $OUT
for i in 1 2 3; do
timeout 600 python bench.py --config base_recipe_hubert --steps 50 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line >> $OUT
done
timeout 600 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line >> $OUT
