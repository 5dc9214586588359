#!/bin/bash
# round-3 GPU trip 2: fused layer-norm backward + attention gradient fusion (mode 1, DPP sums) on hardware; kernel-time
# microbench of the GEMM forms incl. the 64 x 512 full-row probe; bench A/B of the new defaults
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops.py tests/test_attention.py tests/test_engine.py -x -q -m gpu > gpurun_out/t2_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/t2_pytest.log
timeout 900 python tools/gemm_forms_bench.py --rounds 5 > gpurun_out/t2_forms.txt 2>&1
timeout 900 bash tools/run_ab.sh gpurun_out/t2_ab.txt "S2ST_LN_BWD_SPLIT=1" "S2ST_ATTN_GFUSE=0" "S2ST_ATTN_GFUSE=3" "S2ST_LN_BWD_SPLIT=1 S2ST_ATTN_GFUSE=0" > /dev/null 2>&1
tail -5 gpurun_out/t2_pytest.log; cat gpurun_out/t2_forms.txt; cat gpurun_out/t2_ab.txt
