# usage: prof_grep.sh <pattern>  -- rocprofv3 kernel stats of 5 bench steps, rows matching the pattern
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_g -o run -- python3 bench.py --no-other-configs --steps 5 --warmup 2 --cpu-seconds 0 --no-roofline > gpurun_out/prof_g.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_g/run_results.db 7 | grep -E "$1|all kernels"
python3 tools/prof_queues.py gpurun_out/prof_g/run_results.db | grep -E "^queue|step wall"
rm -rf gpurun_out/prof_g
