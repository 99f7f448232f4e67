#!/bin/bash
# Build the working-tree kernels with extra compiler flags into realvsr_amd/csrc/librealvsr_<name>.so (ablations, timelines):
#   tools/build_variant.sh abl1 -DRVSR_ABL=1 ;  RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_abl1.so python tools/dcn_micro.py ...
set -e
NAME=$1; shift
T=$(mktemp -d)
cp realvsr_amd/csrc/*.hip realvsr_amd/csrc/*.h realvsr_amd/csrc/*.inc realvsr_amd/csrc/Makefile "$T"/
make -s -C "$T" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $*" > /dev/null
cp "$T/librealvsr_hip.so" realvsr_amd/csrc/librealvsr_$NAME.so
rm -rf "$T"
echo built librealvsr_$NAME.so
