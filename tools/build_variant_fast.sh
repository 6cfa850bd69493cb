#!/bin/bash
# tools/build_variant_fast.sh <name> "<file1.hip file2.hip ...>" <extra hipcc flags...>: like build_variant.sh, but only the named
# sources are recompiled with the extra flags; every other object is taken from the shipped build (mem_amd/csrc/_build, made
# current by `make` first).  Output: mem_amd/exp/<name>.so (select with MEMHIP_LIB=mem_amd/exp/<name>.so).
set -e
name=$1; files=$2; shift 2
cd "$(dirname "$0")/../mem_amd/csrc"
mkdir -p _build_$name ../exp
objs=""
for o in _build/*.o; do
  b=$(basename $o .o); skip=0
  for f in $files; do [ "$b" = "$f" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in $files; do
  extra=""; case $f in augment.hip|raster.hip|event_norm.hip|records.hip) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -w $extra "$@" -c $f -o _build_$name/$f.o &
done
wait
for f in $files; do objs="$objs _build_$name/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/$name.so $objs
echo built ../exp/$name.so
