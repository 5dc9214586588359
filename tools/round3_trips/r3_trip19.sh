#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 1200 bash tools/attn_stamp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t19_attn_stamp.txt
echo DONE
