#!/bin/bash
# tools/make_variant.sh <name> "<-D flags>" <unit> [<unit> ...]: an A/B build of libpinot_gpu.so with SOME units compiled under other
# compile-time settings, linked with the current objects of all the others -> tools/libpinot_gpu_<name>.so (PINOT_GPU_LIB selects it).
# (`make variant` of pinot_amd/csrc/Makefile does the same for one unit.)
set -e
cd "$(dirname "$0")/../pinot_amd/csrc"
name=$1; defs=$2; shift 2
units=$(sed -n 's/^GPU_UNITS := //p' Makefile)
mkdir -p build/variant_$name
objs=""
for u in $units; do
  if [[ " $* " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $defs -c -o build/variant_$name/$u.o $u.hip 2>/dev/null &
    objs="$objs build/variant_$name/$u.o"
  else
    objs="$objs build/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o ../../tools/libpinot_gpu_$name.so $objs
rm -rf build/variant_$name
ls -la ../../tools/libpinot_gpu_$name.so
