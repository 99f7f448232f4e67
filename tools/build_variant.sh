#!/bin/bash
# Scratch build of ONE kernel file with extra -D flags into realvsr_amd/csrc/librealvsr_<name>.so (load with RVSR_SO=...):
#   tools/build_variant.sh dcn4_kernels abl1 -DRVSR_ABL4=1
set -e
STEM="$1"; NAME="$2"; shift; shift
T=$(mktemp -d)
cp realvsr_amd/csrc/*.hip realvsr_amd/csrc/*.h realvsr_amd/csrc/*.inc realvsr_amd/csrc/Makefile "$T"/
cp realvsr_amd/csrc/*.o "$T"/ 2>/dev/null || true
rm -f "$T"/$STEM.o "$T"/librealvsr_hip.so
make -s -C "$T" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function $*" > /dev/null
cp "$T/librealvsr_hip.so" realvsr_amd/csrc/librealvsr_$NAME.so
rm -rf "$T"
echo built librealvsr_$NAME.so
