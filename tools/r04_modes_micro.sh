#!/bin/bash
# conv / DCN micro-benchmarks in the GEMM modes
for m in bf16x3 bf16x2 bf16; do
  echo "== $m"
  RVSR_GEMM=$m RVSR_MICRO_CHECK=1 timeout 120 python tools/conv_micro.py --iters 20 2>&1 | tail -2
  RVSR_GEMM=$m timeout 120 python tools/conv_micro.py --iters 20 --bwd 2>&1 | tail -1
  RVSR_GEMM=$m timeout 120 python tools/dcn_micro.py --B 40 --iters 40 --ostd 1.25 2>&1 | tail -1
done
