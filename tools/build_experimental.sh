#!/bin/bash
# The product library WITH the measured-and-not-chosen forms (-DS2ST_EXPERIMENTAL: persistent tile walk, stream-K, 256 x 128
# tiles, S2ST_TIMING_SKIP_WGRAD) -> speech-to-speech-translation_amd/csrc/libs2st_hip_experimental.so.  The A/B tools that
# flip those switches load it through S2ST_HIP_LIB=<that path>.  The default build (__graft_entry__.build) never defines it.
set -e
cd "$(dirname "$0")/.."
CSRC=speech-to-speech-translation_amd/csrc
OUT=$CSRC/build_experimental
mkdir -p $OUT
objs=""
for f in $CSRC/*.hip $CSRC/*.cpp; do
  o=$OUT/$(basename $f).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS2ST_EXPERIMENTAL -I $CSRC -I include -Wno-unused-value -x hip -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $CSRC/libs2st_hip_experimental.so
echo $CSRC/libs2st_hip_experimental.so
