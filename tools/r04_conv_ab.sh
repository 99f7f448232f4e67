#!/bin/bash
# conv micro-benchmark: product library against scratch variants (tools/build_variant.sh conv2_kernels <name> ...): tools/r04_conv_ab.sh <name>...
for m in bf16x3 bf16; do
  for v in hip "$@" hip "$@"; do
    echo -n "$m $v fwd: "; RVSR_GEMM=$m RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/conv_micro.py --iters 30 2>&1 | tail -1
    echo -n "$m $v fwd+bwd: "; RVSR_GEMM=$m RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/conv_micro.py --iters 20 --bwd 2>&1 | tail -1
  done
done
