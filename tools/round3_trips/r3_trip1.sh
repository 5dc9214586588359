#!/bin/bash
# round-3 GPU trip 1: correctness of the 4-wave early-release GEMM form on hardware, per-shape timing, bench A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu -k "w4 or group" > gpurun_out/t1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/t1_pytest.log
timeout 900 python tools/gemm_forms_bench.py > gpurun_out/t1_forms.txt 2>&1
timeout 900 bash tools/run_ab.sh gpurun_out/t1_ab.txt "S2ST_GEMM_W4=1" "S2ST_GEMM_W4=2" "S2ST_GEMM_W4=1 S2ST_W4_NS64=2" > /dev/null 2>&1
tail -5 gpurun_out/t1_pytest.log; cat gpurun_out/t1_forms.txt; cat gpurun_out/t1_ab.txt
