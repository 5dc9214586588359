# Where a ring-GEMM workgroup spends its cycles: a PRIVATE build of the library with -DS2ST_GEMM_STAMP (per-workgroup
# clock stamps into the scratch pointer of the launch), driven by tools/gemm_stamp.py.  Never the product library.
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/stamp && cd /tmp/stamp
SRC=$GRAFT_REPO_ROOT/speech-to-speech-translation_amd/csrc
objs=""
for f in $SRC/*.hip $SRC/*.cpp; do
  o=/tmp/stamp/$(basename $f).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS2ST_GEMM_STAMP -I $SRC -I $GRAFT_REPO_ROOT/include -c $f -o $o 2>/dev/null &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o /tmp/stamp/libs2st_stamp.so
cd $GRAFT_REPO_ROOT
S2ST_HIP_LIB=/tmp/stamp/libs2st_stamp.so python3 tools/gemm_stamp.py
