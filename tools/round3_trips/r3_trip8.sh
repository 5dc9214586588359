#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 300 python tools/debug_lnsplit2.py hip 2>&1 | grep -v amdgpu.ids > gpurun_out/t8_lnsplit2.txt
cat gpurun_out/t8_lnsplit2.txt
echo DONE
