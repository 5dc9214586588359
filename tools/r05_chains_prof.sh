#!/bin/bash
# per-queue timelines of one step with the two utterance-half chains (S2ST_CHAINS=2): the gaps on the critical streams
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export S2ST_CHAINS=2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ch2 -o run -- python3 bench.py --no-other-configs --steps 10 --warmup 3 --cpu-seconds 0 --no-roofline --no-host-fed --no-other-configs > gpurun_out/prof_ch2.log 2>&1
python3 tools/prof_queues.py gpurun_out/prof_ch2/run_results.db > gpurun_out/r05_chains_stream_timelines.txt 2>&1
python3 tools/prof_summary.py gpurun_out/prof_ch2/run_results.db 13 > gpurun_out/r05_chains_kernel_stats.txt
rm -rf gpurun_out/prof_ch2
