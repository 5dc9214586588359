#!/bin/bash
# round 3, trip 11: loss sums folded by the finalize kernel, one-pass halo images; whole suite (+ engine tests with a
# NaN-poisoned workspace), bench
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t11_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t11_pytest.log
grep -E "passed|failed|FAILED" gpurun_out/t11_pytest.log | tail -8
S2ST_POISON_WORKSPACE=1 timeout 1200 python -m pytest tests/test_engine.py tests/test_full_size.py tests/test_t2s.py -q -m gpu 2>&1 | tail -4
for i in 1 2; do S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --cpu-seconds 0 > gpurun_out/t11_bench_line.txt 2> gpurun_out/t11_bench_verbose.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/t11_bench_line.txt | head -1; done
grep -E "mel_loss|ls_ce|loss_final|halo|cast_bf16|GPU time on|launches$" gpurun_out/t11_bench_verbose.txt | head -12
echo DONE
