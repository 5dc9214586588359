#!/bin/bash
# round-3 GPU trip 5: ordered sums / new CTC kernel / speaker / t2s CTC on hardware, determinism at full size, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t5_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/t5_pytest.log
timeout 1200 bash tools/run_ab.sh gpurun_out/t5_ab.txt "S2ST_ORDERED_BIAS_SUMS=0" "S2ST_LN_BWD_SPLIT=1" > /dev/null 2>&1
S2ST_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/t5_bench_line.txt 2> gpurun_out/t5_bench_verbose.txt
grep -n "FAILED\|passed\|failed" gpurun_out/t5_pytest.log | tail -12; cat gpurun_out/t5_ab.txt; grep "ctc_kernel\|embed_bwd\|fold_batched" gpurun_out/t5_bench_verbose.txt
