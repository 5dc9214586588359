#!/bin/bash
# round 3, trip 12: D = rowsum(dO * O) inside the attention backward kernels
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -k "attention or engine or full_size or hubert_train or base_size" > gpurun_out/t12_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t12_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t12_pytest.log | tail -8
timeout 1200 bash tools/run_ab.sh gpurun_out/t12_ab.txt "S2ST_ATTN_DVEC_KERNEL=1" > /dev/null 2>&1
cat gpurun_out/t12_ab.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 --steps 40 > gpurun_out/t12_bench_line.txt 2> gpurun_out/t12_bench_verbose.txt
grep -E "flash_bwd|attn_dvec|GPU time on" gpurun_out/t12_bench_verbose.txt | head
echo DONE
