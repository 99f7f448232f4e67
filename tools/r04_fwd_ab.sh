#!/bin/bash
# A/B of the DCN forward generations inside one gpurun call (L1 shape, B = 40): RVSR_DCN_FWD=3 vs 4
mkdir -p gpurun_out
for ostd in ${OSTD:-0.125 1.25}; do
  for gen in 3 4 3 4; do
    echo -n "ostd $ostd gen $gen: "
    RVSR_DCN_FWD=$gen timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd $ostd 2>&1 | tail -1
  done
done
