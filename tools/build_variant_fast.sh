#!/bin/bash
# tools/build_variant_fast.sh <name> "<file1.hip file2.hip ...>" <extra hipcc flags...>: like build_variant.sh, but only the named
# sources (and core.cpp, which records the flags) are compiled with the extra flags; every other object is a copy of the shipped
# build's (mem_amd/csrc/_build, made current by `make` first).  Output: variants/<name>.so.
set -e
name=$1; files=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$root/mem_amd/csrc" -j8 > /dev/null 2>&1
obj="$root/variants/_obj_$name"
mkdir -p "$obj"
cp -p "$root"/mem_amd/csrc/_build/*.o "$obj/"
for f in $files core.cpp; do rm -f "$obj/$f.o"; done
# objects copied with their time stamps are newer than every source: make rebuilds exactly the deleted ones
make -C "$root/mem_amd/csrc" -j8 BUILD="$obj" OUT="$root/variants/$name.so" EXTRA="$*" 2>&1 | grep -E "error|rror:" || true
ls -la "$root/variants/$name.so"
