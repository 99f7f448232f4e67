#!/bin/bash
# conv micro-benchmark A/B of scratch variants (tools/build_variant.sh conv2_kernels <name> ...) against the product library, interleaved, two rounds
for r in 1 2; do
for v in hip "$@"; do
  echo -n "$v fwd: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/conv_micro.py --iters 30 2>&1 | tail -1
  echo -n "$v fwd+bwd: "; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/conv_micro.py --iters 20 --bwd 2>&1 | tail -1
done
done
