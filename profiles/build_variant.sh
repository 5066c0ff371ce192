#!/bin/bash
# A/B builds of one kernel translation unit: build/libsga_<name>.so = the in-tree objects with <unit>.hip
# recompiled under extra flags; select at run time with SGA_LIBRARY_PATH (same box, same run).
#   bash profiles/build_variant.sh <name> <unit> "<extra flags>"
set -e
cd "$(dirname "$0")/../spin-glass-anneal-rl_amd/csrc"
name=$1; unit=$2; extra=$3
mkdir -p ../../build
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -Wall -Wno-unused-function $extra -c $unit.hip -o ../../build/${unit}_${name}.o
objs=$(ls *.o | grep -v "^$unit.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libsga_${name}.so $objs ../../build/${unit}_${name}.o
echo built build/libsga_${name}.so
