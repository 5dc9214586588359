#!/bin/bash
# GPU-box helper: gpu tests (optional), bench line, rocprofv3 kernel stats of the bench step.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r1x}
if [ "${2:-}" = "tests" ]; then timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -4; fi
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | grep -v amdgpu | tail -1 | cut -c1-1500
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run -- python3 bench.py --no-other-configs --steps 5 --warmup 2 --cpu-seconds 0 --no-roofline > gpurun_out/prof_$TAG.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_$TAG/run_results.db 7 > gpurun_out/prof_${TAG}_kernel_stats.txt
head -12 gpurun_out/prof_${TAG}_kernel_stats.txt; python3 tools/prof_queues.py gpurun_out/prof_$TAG/run_results.db
