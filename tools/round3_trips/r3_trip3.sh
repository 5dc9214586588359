#!/bin/bash
# round-3 GPU trip 3: the whole GPU suite, the bench line (verbose), config 5 bench, GEMM forms with the automatic pick
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t3_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/t3_pytest.log
S2ST_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/t3_bench_line.txt 2> gpurun_out/t3_bench_verbose.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --config infer_base > gpurun_out/t3_infer_line.txt 2> gpurun_out/t3_infer_verbose.txt
timeout 900 python tools/gemm_forms_bench.py --rounds 5 > gpurun_out/t3_forms.txt 2>&1
tail -8 gpurun_out/t3_pytest.log; cat gpurun_out/t3_bench_line.txt; cat gpurun_out/t3_infer_line.txt; tail -5 gpurun_out/t3_infer_verbose.txt
