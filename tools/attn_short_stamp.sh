cd $GRAFT_REPO_ROOT
mkdir -p /tmp/astamp
SRC=$GRAFT_REPO_ROOT/speech-to-speech-translation_amd/csrc
objs=""
for f in $SRC/*.hip $SRC/*.cpp; do
  o=/tmp/astamp/$(basename $f).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS2ST_ATTN_STAMP -I $SRC -I $GRAFT_REPO_ROOT/include -Wno-unused-value -c $f -o $o 2>/dev/null &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o /tmp/astamp/libs2st_astamp.so
echo "# dropout 0.1"; S2ST_HIP_LIB=/tmp/astamp/libs2st_astamp.so python3 tools/attn_short_stamp.py
echo "# dropout 0"; DROP_P=0 S2ST_HIP_LIB=/tmp/astamp/libs2st_astamp.so python3 tools/attn_short_stamp.py
