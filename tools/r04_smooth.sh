#!/bin/bash
# DCN pack at the L1 shape (B = 40): i.i.d. offsets against spatially smooth offset fields of the same magnitude
for ostd in 1.25 3.0; do
  for sm in 0 4 16; do
    echo -n "ostd $ostd smooth $sm fwd: "; timeout 120 python tools/dcn_micro.py --B 40 --iters 20 --fwd-only --ostd $ostd --smooth $sm 2>&1 | tail -1
    echo -n "ostd $ostd smooth $sm fwd+bwd: "; timeout 120 python tools/dcn_micro.py --B 40 --iters 10 --ostd $ostd --smooth $sm 2>&1 | tail -1
  done
done
