#!/bin/bash
# fused DCN forward, fixed tile halos (RVSR_DCN3_HALO: 3 / 7 on the 8-row tile, 107 = 7 px on the 16-row tile) against offset scales
for ostd in ${OSTDS:-0.3 0.75 1.25 2.0 3.0}; do
  for h in ${HALOS:-3 7 107}; do
    echo -n "ostd $ostd halo $h: "; RVSR_DCN3_HALO=$h python tools/dcn_micro.py --B 40 --iters 20 --ostd $ostd --fwd-only 2>&1 | grep "dcn pack"
  done
done
