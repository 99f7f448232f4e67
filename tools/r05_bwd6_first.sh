#!/bin/bash
# round 5: first run of the fused DCN backward (dcn6): parity tests, then per-kernel timings against the round-4 pair
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
L=gpurun_out/r05_bwd6_first.log
: > $L
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn_fullsize.py -x -q -m gpu 2>&1 | tail -15 >> $L
for gen in 6 7; do
  for ostd in 0.125 1.25; do
    echo "== RVSR_DCN_BWD=$gen ostd=$ostd" >> $L
    RVSR_DCN_BWD=$gen timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b6_${gen}_${ostd} --output-format csv -- python tools/dcn_micro.py --B 40 --iters 6 --ostd $ostd >> $L 2>&1
    f=$(ls gpurun_out/prof_b6_${gen}_${ostd}/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && head -8 "$f" | cut -c1-200 >> $L
  done
done
tail -60 $L
