#!/bin/bash
OUT=gpurun_out/r05_hubert_gelu.txt
: > $OUT
timeout 1200 python -m pytest tests/test_hubert.py tests/test_gemm.py tests/test_hubert_train.py -q -m gpu 2>&1 | tail -3 >> $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; }
python tools/hubert_timeline.py 2>&1 | sed -n 10,20p >> $OUT
python tools/hubert_timeline.py 2>&1 | tail -1 >> $OUT
S2ST_POSCONV_EACH=1 python tools/hubert_timeline.py 2>&1 | tail -1 >> $OUT
for i in 1 2 3; do
timeout 600 python bench.py --config base_recipe_hubert --steps 50 --warmup 5 --cpu-seconds 0 --no-host-fed --no-roofline --no-other-configs 2>&1 | tail -1 | line >> $OUT
done
