#!/bin/bash
# round 3, trip 10: whole GPU suite, the profile round (r03_f), one-step timeline
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t10_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t10_pytest.log
grep -E "passed|failed|FAILED" gpurun_out/t10_pytest.log | tail -8
timeout 2400 bash tools/profile_round.sh r03_f > gpurun_out/t10_profile_round.log 2>&1
tail -20 gpurun_out/t10_profile_round.log
timeout 300 python bench.py --steps 10 --warmup 5 --cpu-seconds 0 --no-roofline --timeline gpurun_out/r03_f_timeline.txt > /dev/null 2>&1
python tools/timeline.py gpurun_out/r03_f_timeline.txt 8 > gpurun_out/r03_f_timeline_summary_all_dispatches.txt 2>&1
head -40 gpurun_out/r03_f_timeline_summary_all_dispatches.txt
echo DONE
