#!/bin/bash
# Build the committed (HEAD) kernels into realvsr_amd/csrc/librealvsr_head.so so that the working-tree build can be
# A/B-compared against it inside ONE gpurun call (box-to-box spread is ~5 %, larger than most single changes):
#   RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_head.so python bench.py ...   vs   python bench.py ...
set -e
T=$(mktemp -d)
git -C "$(dirname "$0")/.." archive ${1:-HEAD} realvsr_amd/csrc | tar -x -C "$T"
make -s -C "$T/realvsr_amd/csrc" -j8 > /dev/null
cp "$T/realvsr_amd/csrc/librealvsr_hip.so" "$(dirname "$0")/../realvsr_amd/csrc/librealvsr_head.so"
rm -rf "$T"
echo built librealvsr_head.so from $(git -C "$(dirname "$0")/.." rev-parse --short ${1:-HEAD})
