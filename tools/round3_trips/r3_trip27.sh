#!/bin/bash
# round 3, trip 27: smoke() as the driver runs it, the GPU suite a second and third time (flakiness), default bench.py
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
for i in 1 2; do timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2; done
timeout 900 python bench.py 2>/dev/null | cut -c1-200
echo DONE
