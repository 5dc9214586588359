cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export S2ST_NO_SIDE_STREAM=1
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ns -o run -- python3 bench.py --no-other-configs --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline > gpurun_out/prof_ns.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_ns/run_results.db 25 > gpurun_out/r02_d_single_stream_kernel_stats.txt
head -3 gpurun_out/r02_d_single_stream_kernel_stats.txt
rm -rf gpurun_out/prof_ns
