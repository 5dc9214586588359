#!/bin/bash
# round 3, trip 20: 32-bit dropout mixer (two 32-bit multiplies instead of three 64-bit ones)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t20_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t20_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t20_pytest.log | tail -8
for i in 1 2; do S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t20_bench_line.txt 2> gpurun_out/t20_bench_verbose.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/t20_bench_line.txt | head -1; done
grep -E "flash_|GPU time on|layernorm_bwd_fused|gemm_bf16_dma_kernel<128, 64, true, true|w4_kernel" gpurun_out/t20_bench_verbose.txt | head
timeout 1200 bash tools/attn_stamp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t20_attn_stamp.txt
echo DONE
