#!/bin/bash
# GPU-box helper: everything profiles/<tag>_* is made of.  usage: profile_round.sh <tag> [bench args...]
#   1. the bench line (un-profiled)               2. rocprofv3 --kernel-trace --stats of the same command
#   3. per-stream timelines of one step           4. PMC passes (own runs, kernel trace only): FETCH_SIZE, WRITE_SIZE,
#      SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE  5. per-kernel HBM traffic JSON (tools/pmc_traffic.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out
CFG=base_recipe
for a in "$@"; do if [ "$prev" = "--config" ]; then CFG=$a; fi; prev=$a; done
S2ST_BENCH_VERBOSE=1 python3 bench.py --steps 20 --warmup 5 "$@" > $O/${TAG}_bench_verbose.txt 2>&1
grep '^{' $O/${TAG}_bench_verbose.txt > $O/${TAG}_bench_line.txt
# steps the profiled command executes: training configs 5 warm-up + 20 timed + 20 replayed (the host-fed leg is switched off
# under the profiler); infer_base 5 + 20 timed + 20 with the device's phase generator + decode / vocoder split + roofline pass
NSTEPS=45; HF=--no-host-fed
if [ "$CFG" = "infer_base" ]; then NSTEPS=48; HF=; fi
rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o run -- python3 bench.py --no-other-configs --steps 20 --warmup 5 --cpu-seconds 0 $HF "$@" > $O/prof_$TAG.log 2>&1
grep '^{' $O/prof_$TAG.log > $O/${TAG}_bench_line_under_rocprof.txt
python3 tools/prof_summary.py $O/prof_$TAG/run_results.db $NSTEPS > $O/${TAG}_kernel_stats.txt
python3 tools/prof_summary.py $O/prof_$TAG/run_results.db 20 last 20 > $O/${TAG}_kernel_stats_replayed_steps.txt   # the launches behind the bench line's roofline
python3 tools/prof_queues.py $O/prof_$TAG/run_results.db > $O/${TAG}_stream_timelines.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $O/pmc_${TAG}_$C -o run -- python3 bench.py --no-other-configs --steps 3 --warmup 2 --cpu-seconds 0 --no-roofline $HF "$@" > $O/pmc_${TAG}_$C.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_${TAG}_$C/run_results.db > $O/${TAG}_pmc_$(echo $C | tr A-Z a-z).txt 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_${TAG}_MFMA -o run -- python3 bench.py --no-other-configs --steps 3 --warmup 2 --cpu-seconds 0 --no-roofline $HF "$@" > $O/pmc_${TAG}_MFMA.log 2>&1
python3 tools/pmc_summary.py $O/pmc_${TAG}_MFMA/run_results.db > $O/${TAG}_pmc_mfma_busy.txt 2>&1
python3 tools/pmc_traffic.py $O/pmc_${TAG}_FETCH_SIZE/run_results.db $O/pmc_${TAG}_WRITE_SIZE/run_results.db $CFG $O/${TAG}_pmc_traffic.json 5
rm -rf $O/prof_$TAG $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE $O/pmc_${TAG}_MFMA   # raw traces stay on the box (64 MiB pull limit)
head -14 $O/${TAG}_kernel_stats.txt; cat $O/${TAG}_bench_line.txt | cut -c1-600
