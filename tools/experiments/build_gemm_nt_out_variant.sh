#!/bin/bash
# Variant library: GEMM epilogues store a bf16 result larger than 128 MB nontemporally (-DS2ST_GEMM_NT_OUT=128) -> csrc/libs2st_hip_ntout.so
set -e
cd "$(dirname "$0")/../.."
CSRC=speech-to-speech-translation_amd/csrc
python -c "import __graft_entry__ as g; g.build()"
OUT=$CSRC/build_ntout
mkdir -p $OUT
for f in gemm_bf16.hip gemm_bf16_w4.hip gemm_bf16_p4.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS2ST_GEMM_NT_OUT=128 -I $CSRC -I include -Wno-unused-value -x hip -c $CSRC/$f -o $OUT/$f.o &
done
wait
objs=""
for o in $CSRC/build/*.o; do
  b=$(basename $o)
  if [ -f $OUT/$b ]; then objs="$objs $OUT/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $CSRC/libs2st_hip_ntout.so
echo $CSRC/libs2st_hip_ntout.so
