#!/bin/bash
# round 3, trip 28: the GPU suite twice more (flakiness), default bench.py line saved in full
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t28_pytest_$i.log 2>&1; echo "run $i rc $?"; grep -E "passed|failed" gpurun_out/t28_pytest_$i.log | tail -1; done
timeout 900 python bench.py > gpurun_out/t28_bench_line.txt 2>/dev/null
python - <<'PY'
import json
l = json.loads(open("gpurun_out/t28_bench_line.txt").read())
print(l["value"], l["ms_per_step"], "traffic", l["roofline"]["traffic"], l["roofline"]["traffic_source"], "cpu", l["cpu_baseline"]["value"])
PY
echo DONE
