#!/bin/bash
# round 3, trip 25: optimizer update overlapped with the next forward
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t25_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t25_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t25_pytest.log | tail -8
timeout 1200 bash tools/run_ab.sh gpurun_out/t25_ab.txt "S2ST_ADAM_OVERLAP=0" "S2ST_ADAM_CHUNKS=4" "S2ST_ADAM_CHUNKS=16" > /dev/null 2>&1
cat gpurun_out/t25_ab.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t25_bench_line.txt 2> gpurun_out/t25_bench_verbose.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/t25_bench_line.txt | head -1
grep -E "adam|GPU time on|per-step GPU" gpurun_out/t25_bench_verbose.txt | head -6
echo DONE
