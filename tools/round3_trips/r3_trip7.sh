#!/bin/bash
# round 3, trip 7: where the fused and the split layer-norm backward part ways on hardware; ordered embedding backward
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for v in "S2ST_NO_SIDE_STREAM=1" "S2ST_NO_LN_FUSE=1" "S2ST_JOIN_EVERY_SEGMENT=1" "S2ST_NO_WGRAD_GROUP=1" "S2ST_LN_RPW=1"; do
  timeout 300 python tools/debug_lnsplit.py hip $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/t7_lnsplit.txt
cat gpurun_out/t7_lnsplit.txt
timeout 600 python -m pytest tests -q -m gpu -k "test_elementwise or embed or ops" -x 2>&1 | tail -3
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t7_bench_line.txt 2> gpurun_out/t7_bench_verbose.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/t7_bench_line.txt | head -1
grep -E "embed_bwd|GPU time on" gpurun_out/t7_bench_verbose.txt | head
echo DONE
