#!/bin/bash
# does keeping the weight-gradient stream off part of the chip help the data path?
#   tools/cu_partition.sh probe     -- which CUs a masked stream runs on (tools/ubench/cu_mask_probe.hip)
#   tools/cu_partition.sh bench     -- bench.py under a few partitions, alternating with the unmasked default
# run on the GPU box (gpurun -- 'tools/cu_partition.sh probe > gpurun_out/x.txt 2>&1')
set -u
cd "$(dirname "$0")/.."
case "${1:-probe}" in
probe)
  hipcc --offload-arch=gfx950 -O3 tools/ubench/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
  ;;
bench)
  run() {  # name, env...
    local name="$1"; shift
    local line
    line=$(env "$@" python bench.py --steps 40 --warmup 5 --no-roofline --cpu-seconds 0 2>/dev/null | grep '"metric"')
    echo "$name: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
  }
  # mapping found by the probe (profiles/r04_cu_mask_probe.txt): mask bit i = XCD (i % 8), CU (i / 8) of that XCD -- the
  # low 8 k bits are k CUs on every XCD; an XCD with no bit set gets ALL its CUs
  F=ffffffff
  MASK8="$F,$F,0,0,0,0,0,0";      MASK24C="0,0,$F,$F,$F,$F,$F,$F"
  MASK12="$F,$F,$F,0,0,0,0,0";    MASK20C="0,0,0,$F,$F,$F,$F,$F"
  MASK16="$F,$F,$F,$F,0,0,0,0";   MASK16C="0,0,0,0,$F,$F,$F,$F"
  for rep in 1 2; do
    run "default                         " S2ST_NOP=1
    run "side 8/32 per XCD               " S2ST_SIDE_CU_MASK="$MASK8"
    run "side 8/32, main 24/32 (192 CUs) " S2ST_SIDE_CU_MASK="$MASK8" S2ST_MAIN_CU_MASK="$MASK24C" S2ST_DATA_CUS=192
    run "side 16/32 per XCD              " S2ST_SIDE_CU_MASK="$MASK16"
    run "side 16/32, main 16/32          " S2ST_SIDE_CU_MASK="$MASK16" S2ST_MAIN_CU_MASK="$MASK16C" S2ST_DATA_CUS=128
    run "side 12/32 per XCD              " S2ST_SIDE_CU_MASK="$MASK12"
    run "side 12/32, main 20/32 (160 CUs)" S2ST_SIDE_CU_MASK="$MASK12" S2ST_MAIN_CU_MASK="$MASK20C" S2ST_DATA_CUS=160
    run "main 24/32 only (side unmasked) " S2ST_MAIN_CU_MASK="$MASK24C" S2ST_DATA_CUS=192
  done
  ;;
esac
