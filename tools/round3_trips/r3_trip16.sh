#!/bin/bash
# round 3, trip 16: group-size A/B, then the final profile round (r03_g) + one-step timeline + determinism probe
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 bash tools/run_ab.sh gpurun_out/t16_ab.txt "S2ST_WGRAD_GROUP=4" "S2ST_WGRAD_GROUP=8" "S2ST_WGRAD_GROUP=12" "S2ST_SPLITK_TARGET=256" > /dev/null 2>&1
cat gpurun_out/t16_ab.txt
echo DONE
