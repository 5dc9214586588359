#!/bin/bash
# Build the product's HIP sources for the HOST against the wave64 emulator (tests only).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$(dirname "$HERE")")"
SRC="$ROOT/speech-to-speech-translation_amd/csrc"
OUT="$HERE/_build"
mkdir -p "$OUT"
CXX=/opt/rocm/lib/llvm/bin/clang++
# -DS2ST_EXPERIMENTAL: the emulator build keeps the measured-and-not-chosen GEMM forms (persistent walk, stream-K, 256 x 128)
# that the product library no longer carries, so that their sources stay tested on the CPU
FLAGS="-DS2ST_EXPERIMENTAL -std=c++17 -O2 -g -fPIC -pthread -I$HERE -I$SRC -I$ROOT/include -Wno-unused-value -Wno-vla-cxx-extension"
objs=""
for f in "$SRC"/*.hip "$SRC"/*.cpp "$HERE/emu_runtime.cpp"; do
  [ -e "$f" ] || continue
  o="$OUT/$(basename "$f").o"
  if [ ! -e "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find "$SRC" "$HERE/hip" "$ROOT/include" -name '*.h' -newer "$o" 2>/dev/null | head -1)" ]; then
    $CXX $FLAGS -x c++ -c "$f" -o "$o" &
  fi
  objs="$objs $o"
done
wait
$CXX -shared -pthread $objs -ldl -o "$OUT/libs2st_emu.so"
echo "$OUT/libs2st_emu.so"
