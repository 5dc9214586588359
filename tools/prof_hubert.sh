cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_hub -o run -- python3 tools/hubert_bench.py 24 8 > gpurun_out/prof_hub.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_hub/run_results.db 10 | head -24
rm -rf gpurun_out/prof_hub
