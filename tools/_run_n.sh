python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r04_aa_gputests.txt
python bench.py > gpurun_out/r04_aa_bench.txt 2>gpurun_out/r04_aa_bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_aa_smoke.txt 2>&1
tail -4 gpurun_out/r04_aa_gputests.txt; cut -c1-400 gpurun_out/r04_aa_bench.txt; tail -2 gpurun_out/r04_aa_smoke.txt
