#!/bin/bash
# round 3, trip 36: run-to-run spread of the default bench line on one box, GEMM clock stamps
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
: > gpurun_out/r03_k_run_to_run_spread.txt
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --cpu-seconds 0 --no-roofline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('run %s: %d steps, %.3f ms/step, %.0f mel-frames/s' % (sys.argv[1], l['steps'], l['ms_per_step'], l['value']))" $i | tee -a gpurun_out/r03_k_run_to_run_spread.txt
done
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --cpu-seconds 0 --no-roofline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('run %s: %d steps, %.3f ms/step, %.0f mel-frames/s' % (sys.argv[1], l['steps'], l['ms_per_step'], l['value']))" $i | tee -a gpurun_out/r03_k_run_to_run_spread.txt
done
timeout 1500 bash tools/gemm_stamp.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_k_gemm_stamps.txt
cat gpurun_out/r03_k_gemm_stamps.txt | cut -c1-250
echo DONE
