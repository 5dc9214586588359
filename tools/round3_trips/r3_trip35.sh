#!/bin/bash
# round 3, trip 35 (experiment): grouped weight gradients with three ring stages (96 KB LDS)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -k "group" 2>&1 | tail -2
timeout 1500 bash tools/run_ab.sh gpurun_out/t35_ab.txt "S2ST_GROUP_NS=3" "S2ST_GROUP_NS=3 S2ST_GEMM_W4=1" > /dev/null 2>&1
cat gpurun_out/t35_ab.txt
S2ST_GROUP_NS=3 S2ST_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 40 --cpu-seconds 0 2>&1 >/dev/null | grep -E "group_kernel|w4_kernel<128, 64, true, true|layernorm_bwd_fused_kernel<true>" | head -4
echo DONE
