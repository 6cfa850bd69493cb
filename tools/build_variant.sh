#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: an A/B build of libmemhip.so into mem_amd/exp/<name>.so
# (select it with MEMHIP_LIB=mem_amd/exp/<name>.so); objects under mem_amd/csrc/_build_<name>/
set -e
name=$1; shift
cd "$(dirname "$0")/../mem_amd/csrc"
mkdir -p _build_$name ../exp
for f in core.cpp mask.cpp *.hip; do
  extra=""; case $f in augment.hip|raster.hip|event_norm.hip|records.hip) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math $extra "$@" -c $f -o _build_$name/$f.o &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/$name.so _build_$name/*.o
echo built ../exp/$name.so
