#!/bin/bash
# round 3, trip 6: CTC in three launches, Adam clears the gradient arena, conv0 GroupNorm without atomics
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -k "ctc or adam or hubert or base_size_ar or fused_backward or t2s or full_size" > gpurun_out/t6_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t6_pytest.log
timeout 600 python tools/debug_lnsplit.py hip > gpurun_out/t6_lnsplit.txt 2>&1
tail -45 gpurun_out/t6_lnsplit.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py > gpurun_out/t6_bench_line.txt 2> gpurun_out/t6_bench_verbose.txt
cat gpurun_out/t6_bench_line.txt
grep -E "ctc|adam|Fill|fill|log_softmax" gpurun_out/t6_bench_verbose.txt | head
echo DONE
