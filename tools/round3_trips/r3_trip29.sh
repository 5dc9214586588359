#!/bin/bash
# round 3, trip 29: grouped weight gradients on the 4-wave form (64 KB LDS: co-residency with data-path workgroups)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 1500 bash tools/run_ab.sh gpurun_out/t29_ab.txt "S2ST_GROUP_W4=1" "S2ST_GROUP_W4=1 S2ST_GEMM_W4=1" > /dev/null 2>&1
cat gpurun_out/t29_ab.txt
S2ST_GROUP_W4=1 S2ST_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 40 --cpu-seconds 0 2>&1 >/dev/null | grep -E "group_kernel|flash_bwd|layernorm_bwd_fused_kernel<true>|dma_kernel<128, 64, true, true" | head -6
echo DONE
