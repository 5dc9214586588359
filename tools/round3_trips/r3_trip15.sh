#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 600 python tools/ln_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t15_ln_bench.txt
timeout 900 bash tools/run_ab.sh gpurun_out/t15_ab.txt "S2ST_LN_RPW=1" "S2ST_LN_RPW=2" > /dev/null 2>&1
cat gpurun_out/t15_ab.txt
echo DONE
