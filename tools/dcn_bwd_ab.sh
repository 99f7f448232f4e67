#!/bin/bash
# DCN backward, one gpurun call: parity of the DCN suites, per-kernel A/B of library variants, timeline
#   tools/dcn_bwd_ab.sh [variant specs for tools/ab_kernels.sh ...]
mkdir -p gpurun_out
L=gpurun_out/dcn_bwd_ab.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn_fullsize.py -x -q -m gpu 2>&1 | tail -6 >> $L
bash tools/ab_kernels.sh "$@" >> $L 2>&1
bash tools/ab_kernels.sh -m "--B 16 --C 128 --iters 6 --ostd 1.25" "$@" >> $L 2>&1
if [ -f realvsr_amd/csrc/librealvsr_tl6.so ]; then RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl6.so python tools/dcn6_timeline.py 1.25 2>&1 | grep -v amdgpu.ids >> $L; fi
cat $L
