#!/bin/bash
# round 3, trip 33: final sources -- whole GPU suite, profile round r03_k, workloads, determinism probe
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t26_pytest.log 2>&1
echo "pytest rc $?" | tee -a gpurun_out/t26_pytest.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/t26_pytest.log | tail -8
timeout 2400 bash tools/profile_round.sh r03_k > gpurun_out/t26_profile_round.log 2>&1
tail -14 gpurun_out/t26_profile_round.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 5 --cpu-seconds 0 --no-roofline --timeline gpurun_out/r03_k_timeline.txt > /dev/null 2>&1
python tools/timeline.py gpurun_out/r03_k_timeline.txt 8 > gpurun_out/r03_k_timeline_summary_all_dispatches.txt 2>&1
head -2 gpurun_out/r03_k_timeline_summary_all_dispatches.txt; grep "^stream 1" gpurun_out/r03_k_timeline_summary_all_dispatches.txt
timeout 900 python bench.py > gpurun_out/r03_k_bench_line_100_steps.txt 2> /dev/null
cut -c1-330 gpurun_out/r03_k_bench_line_100_steps.txt
timeout 900 python bench.py --config infer_base > gpurun_out/r03_k_infer_base_line.txt 2> /dev/null
cut -c1-200 gpurun_out/r03_k_infer_base_line.txt
timeout 900 python bench.py --config base_recipe_hubert --cpu-seconds 0 > gpurun_out/r03_k_bench_hubert_line.txt 2> /dev/null
cut -c1-260 gpurun_out/r03_k_bench_hubert_line.txt
timeout 900 bash tools/cold_probe.sh 16 > gpurun_out/r03_k_cold_probe.txt 2>&1
cut -c1-160 gpurun_out/r03_k_cold_probe.txt
S2ST_NO_SIDE_STREAM=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/single stream: /' | tee gpurun_out/r03_k_single_stream.txt
timeout 1200 bash tools/run_ab.sh gpurun_out/r03_k_ab_switches.txt "S2ST_GROUP_XCD=0" "S2ST_ATTN_XCD=0" > /dev/null 2>&1
cat gpurun_out/r03_k_ab_switches.txt
echo DONE
