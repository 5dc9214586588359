#!/bin/bash
# round 3, trip 17: profile round r03_g (bench line, kernel stats, stream timelines, PMC passes, traffic JSON), one-step
# timeline, other workloads' bench lines, determinism across fresh processes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 2400 bash tools/profile_round.sh r03_g > gpurun_out/t17_profile_round.log 2>&1
tail -18 gpurun_out/t17_profile_round.log | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 5 --cpu-seconds 0 --no-roofline --timeline gpurun_out/r03_g_timeline.txt > /dev/null 2>&1
python tools/timeline.py gpurun_out/r03_g_timeline.txt 8 > gpurun_out/r03_g_timeline_summary_all_dispatches.txt 2>&1
head -3 gpurun_out/r03_g_timeline_summary_all_dispatches.txt; grep "^stream 1" gpurun_out/r03_g_timeline_summary_all_dispatches.txt
timeout 900 python bench.py > gpurun_out/r03_g_bench_line_100_steps.txt 2> /dev/null
cut -c1-400 gpurun_out/r03_g_bench_line_100_steps.txt
timeout 900 python bench.py --config infer_base > gpurun_out/r03_g_infer_base_line.txt 2> gpurun_out/r03_g_infer_base_verbose.txt
cut -c1-500 gpurun_out/r03_g_infer_base_line.txt
S2ST_BENCH_VERBOSE=1 timeout 900 python bench.py --config base_recipe_hubert --cpu-seconds 0 > gpurun_out/r03_g_bench_hubert_line.txt 2> gpurun_out/r03_g_bench_hubert_verbose.txt
cut -c1-300 gpurun_out/r03_g_bench_hubert_line.txt
timeout 900 bash tools/cold_probe.sh 24 > gpurun_out/r03_g_cold_probe.txt 2>&1
cat gpurun_out/r03_g_cold_probe.txt | cut -c1-600
S2ST_NO_SIDE_STREAM=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/single stream: /' | tee gpurun_out/r03_g_single_stream.txt
echo DONE
