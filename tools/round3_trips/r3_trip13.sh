#!/bin/bash
# round 3, trip 13: BatchNorm statistics in one pass, BatchNorm output written as the next convolution's bf16 halo image
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t13_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t13_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t13_pytest.log | tail -8
S2ST_POISON_WORKSPACE=1 timeout 1200 python -m pytest tests/test_engine.py tests/test_full_size.py tests/test_t2s.py -q -m gpu 2>&1 | tail -3
timeout 1200 bash tools/run_ab.sh gpurun_out/t13_ab.txt "S2ST_BN_TWO_PASS=1" > /dev/null 2>&1
cat gpurun_out/t13_ab.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t13_bench_line.txt 2> gpurun_out/t13_bench_verbose.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/t13_bench_line.txt | head -1
grep -E "bn_|colreduce|GPU time on|cast_bf16_halo" gpurun_out/t13_bench_verbose.txt | head
echo DONE
