#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 300 python tools/debug_lnsplit3.py hip 2>&1 | grep -v amdgpu.ids > gpurun_out/t9_lnsplit3.txt
cat gpurun_out/t9_lnsplit3.txt | tail -90
echo DONE
