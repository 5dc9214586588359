#!/bin/bash
# round 3, trip 21: GEMM clock stamps on the current sources
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 bash tools/gemm_stamp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t21_gemm_stamp.txt
echo DONE
