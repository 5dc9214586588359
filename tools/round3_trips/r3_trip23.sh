#!/bin/bash
# round 3, trip 23: straight-line element work in the attention kernels
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python -m pytest tests -q -m gpu -k "attention or engine or full_size or hubert_train or base_size or inference" > gpurun_out/t23_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t23_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t23_pytest.log | tail -8
for i in 1 2; do S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t23_bench_line.txt 2> gpurun_out/t23_bench_verbose.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/t23_bench_line.txt | head -1; done
grep -E "flash_|GPU time on" gpurun_out/t23_bench_verbose.txt | head
timeout 1200 bash tools/attn_stamp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t23_attn_stamp.txt
echo DONE
