#!/bin/bash
# TEST INFRASTRUCTURE.  Builds the one natively compiled piece of the reference on this path --
# fairseq/data/data_utils_fast.pyx (the max-tokens batcher, a Cython extension the reference's setup.py:75-86
# compiles) -- from the source where it lies under /root/reference, with outputs only into oracle/_ref/
# (git-ignored).  No reference source is copied into the repo.  Needs Cython + g++ + numpy headers (present here).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC=/root/reference/fairseq/data/data_utils_fast.pyx
OUT="$HERE/_ref"
[ -f "$SRC" ] || { echo "reference not present: nothing to build"; exit 0; }
mkdir -p "$OUT"
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
if [ "$OUT/data_utils_fast$EXT" -nt "$SRC" ]; then exit 0; fi
python3 -m cython -3 --cplus "$SRC" -o "$OUT/data_utils_fast.cpp"
g++ -O2 -shared -fPIC -std=c++17 -w \
  $(python3 -c "import sysconfig, numpy; print('-I' + sysconfig.get_paths()['include'], '-I' + numpy.get_include())") \
  "$OUT/data_utils_fast.cpp" -o "$OUT/data_utils_fast$EXT"
rm -f "$OUT/data_utils_fast.cpp"
echo "built $OUT/data_utils_fast$EXT"
