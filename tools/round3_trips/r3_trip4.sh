#!/bin/bash
# round-3 GPU trip 4: whole GPU suite; bench A/B of the new switches; config 5 bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t4_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/t4_pytest.log
timeout 1200 bash tools/run_ab.sh gpurun_out/t4_ab.txt "S2ST_LN_RPW=2" "S2ST_WGRAD_TILES=0" "S2ST_GEMM_W4=0" "S2ST_GEMM_W4=2" > /dev/null 2>&1
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --config infer_base > gpurun_out/t4_infer_line.txt 2> gpurun_out/t4_infer_verbose.txt
tail -6 gpurun_out/t4_pytest.log; cat gpurun_out/t4_ab.txt; cut -c1-400 gpurun_out/t4_infer_line.txt
