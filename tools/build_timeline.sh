#!/bin/bash
# Build the working-tree kernels with the s_memtime stamps enabled into realvsr_amd/csrc/librealvsr_tl.so
# (RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl.so python tools/dcn3_timeline.py)
set -e
T=$(mktemp -d)
cp realvsr_amd/csrc/*.hip realvsr_amd/csrc/*.h realvsr_amd/csrc/*.inc realvsr_amd/csrc/Makefile "$T"/
make -s -C "$T" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function ${TLFLAGS:--DRVSR_TIMELINE_DCN}" > /dev/null
cp "$T/librealvsr_hip.so" realvsr_amd/csrc/librealvsr_tl.so
rm -rf "$T"
echo built librealvsr_tl.so
