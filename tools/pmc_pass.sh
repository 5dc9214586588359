#!/bin/bash
# GPU-box helper: one rocprofv3 PMC pass over the bench step (counters in their own run, no tracing
# domains besides kernel dispatch), then a per-kernel summary.  usage: pmc_pass.sh <tag> <counter...>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/pmc_$TAG -o run -- python3 bench.py --no-other-configs --steps 3 --warmup 2 --cpu-seconds 0 --no-roofline > gpurun_out/pmc_$TAG.log 2>&1
ls gpurun_out/pmc_$TAG | head
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG/run_results.db > gpurun_out/pmc_${TAG}_summary.txt 2>&1
head -30 gpurun_out/pmc_${TAG}_summary.txt
