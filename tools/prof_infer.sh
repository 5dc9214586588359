cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_inf -o run -- python3 tools/infer_bench.py 16 100 > gpurun_out/prof_inf.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_inf/run_results.db 1 | head -${1:-24}
rm -rf gpurun_out/prof_inf
