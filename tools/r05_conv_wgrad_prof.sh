#!/bin/bash
# kernel stats + stream timelines of the step with / without the grouped convolution weight gradients
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  export S2ST_CONV_WGRAD_GROUP=$((2 - 2 * v))   # (v = 0: all convolutions grouped; v = 1: the split-row products)
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cw$v -o run -- python3 bench.py --no-other-configs --steps 10 --warmup 3 --cpu-seconds 0 --no-roofline --no-host-fed --no-other-configs > gpurun_out/prof_cw$v.log 2>&1
  python3 tools/prof_summary.py gpurun_out/prof_cw$v/run_results.db 13 > gpurun_out/r05_conv_wgrad_kernel_stats_nogroup$v.txt
  python3 tools/prof_queues.py gpurun_out/prof_cw$v/run_results.db > gpurun_out/r05_conv_wgrad_timelines_nogroup$v.txt 2>&1
  rm -rf gpurun_out/prof_cw$v
done
