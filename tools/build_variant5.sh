#!/bin/bash
# Fast ablation build: only dcn5_kernels.hip is recompiled with extra flags and linked against the product objects:
#   tools/build_variant5.sh a1 -DRVSR_ABL5=1   ->  realvsr_amd/csrc/librealvsr_a1.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../realvsr_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c dcn5_kernels.hip -o /tmp/dcn5_$NAME.o
OBJS=$(ls *.o | grep -v dcn5_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dcn5_$NAME.o -o librealvsr_$NAME.so
echo built librealvsr_$NAME.so
