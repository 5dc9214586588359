#!/bin/bash
# round 3, trip 14: first convolution's weight gradient on the idle data-path stream; single-barrier layer-norm fold
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -k "engine or full_size or layernorm or hubert_train or distributed" > gpurun_out/t14_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t14_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t14_pytest.log | tail -8
timeout 1200 bash tools/run_ab.sh gpurun_out/t14_ab.txt "S2ST_CONV_TAIL_MAIN=0" > /dev/null 2>&1
cat gpurun_out/t14_ab.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t14_bench_line.txt 2> gpurun_out/t14_bench_verbose.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/t14_bench_line.txt | head -1
grep -E "layernorm_bwd|GPU time on" gpurun_out/t14_bench_verbose.txt | head
echo DONE
