#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: a measurement build of libmemhip.so -> variants/<name>.so (select it with
# MEMHIP_LIB=variants/<name>.so; bench.py prints the path and the flags of the library it measured and refuses the headline for
# anything but the shipped build).  Objects under variants/_obj_<name>/ (not shipped to the GPU box: .gpurunignore).  variants/ is
# git-ignored: delete it when the measurement is done.  Experiments whose switches are no longer in the sources are patches under
# tools/exp/ (r05_lab_switches.patch restores every round-5 switch): apply, build, revert.
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/variants"
make -C "$root/mem_amd/csrc" -j8 BUILD="$root/variants/_obj_$name" OUT="$root/variants/$name.so" EXTRA="$*" 2>&1 | grep -E "error|Error|built|rror:" || true
ls -la "$root/variants/$name.so"
