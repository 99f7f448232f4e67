#!/bin/bash
# Fast ablation build: only dcn6_kernels.hip is recompiled with extra flags and linked against the product objects:
#   tools/build_variant6.sh a1 -DRVSR_ABL6=1   ->  realvsr_amd/csrc/librealvsr_a1.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../realvsr_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c dcn6_kernels.hip -o /tmp/dcn6_$NAME.o
OBJS=$(ls *.o | grep -v dcn6_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dcn6_$NAME.o -o librealvsr_$NAME.so
echo built librealvsr_$NAME.so
