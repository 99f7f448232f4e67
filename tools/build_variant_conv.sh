#!/bin/bash
# Fast experiment build: only conv2_kernels.hip is recompiled with extra flags and linked against the product objects:
#   tools/build_variant_conv.sh c1 -DRVSR_ABLC=1   ->  realvsr_amd/csrc/librealvsr_c1.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../realvsr_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c conv2_kernels.hip -o /tmp/conv2_$NAME.o
OBJS=$(ls *.o | grep -v conv2_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/conv2_$NAME.o -o librealvsr_$NAME.so
echo built librealvsr_$NAME.so
