#!/bin/bash
# round 3, trip 32: attention kernels with an XCD-aware block order (A/B, kernel times, PMC fetch)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests -q -m gpu -k "attention or engine or full_size or inference" 2>&1 | tail -2
timeout 1500 bash tools/run_ab.sh gpurun_out/t32_ab.txt "S2ST_ATTN_XCD=0" > /dev/null 2>&1
cat gpurun_out/t32_ab.txt
for v in 1 0; do S2ST_ATTN_XCD=$v S2ST_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 40 --cpu-seconds 0 2>&1 >/dev/null | grep -E "flash_" | head -2; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
  S2ST_ATTN_XCD=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_t32_$v -o run -- python3 bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-roofline > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_t32_$v/run_results.db 2>&1 | grep -E "flash_" | head -2
  rm -rf gpurun_out/pmc_t32_$v
done
echo DONE
