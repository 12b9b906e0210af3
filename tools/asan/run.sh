#!/bin/bash
# AddressSanitizer / UBSan pass over the C++ host mirror without Python (an LD_PRELOADed libasan cannot intercept __cxa_throw there):
#   tools/asan/run.sh [segment directories...]
# builds the two drivers against pinot_amd/csrc/host/*.cpp and runs the SQL parser / predicate lowering cases and the
# segment-directory loader (device -1: nothing is opened on a GPU) over the directories given (tests/segment_dirs.py writes some).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/pinot_asan
mkdir -p "$OUT"
FLAGS="-O0 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -pthread"
# the host sources are compiled once, in parallel, and linked into both drivers
OBJS=""
for src in "$ROOT"/pinot_amd/csrc/host/*.cpp "$ROOT"/tools/asan/parser_driver.cpp "$ROOT"/tools/asan/loader_driver.cpp; do
  obj="$OUT/$(basename "$src" .cpp).o"
  g++ $FLAGS -c -o "$obj" "$src" &
  case "$src" in *_driver.cpp) ;; *) OBJS="$OBJS $obj" ;; esac
done
wait
g++ $FLAGS -o "$OUT/parser_driver" "$OUT/parser_driver.o" $OBJS -ldl
g++ $FLAGS -o "$OUT/loader_driver" "$OUT/loader_driver.o" $OBJS -ldl
ASAN_OPTIONS=detect_leaks=1 "$OUT/parser_driver"
if [ $# -gt 0 ]; then ASAN_OPTIONS=detect_leaks=1 "$OUT/loader_driver" "$@"; fi
echo "asan: clean"

# The C oracle under the sanitizers (it is C: an LD_PRELOADed libasan works with Python):
#   gcc -O1 -g -fsanitize=address,undefined -fPIC -shared -o oracle/_build/libpinot_oracle.so oracle/pinot_oracle.c -lm
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan/run_oracle_tests.py
#   (cd oracle && make clean && make)      # back to the optimised build
