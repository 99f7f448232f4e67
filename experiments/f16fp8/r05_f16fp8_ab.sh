#!/bin/bash
# f16 + fp8 experiment build (tools/build_variant.sh conv2_kernels f16fp8 -DRVSR_F16FP8; forward 3x3 convs only) against the product
export RVSR_PACK_CACHE=0
for r in 1 2; do for v in hip f16fp8; do
  echo "== $v"; RVSR_MICRO_CHECK=1 RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 120 python tools/conv_micro.py --iters 200 2>&1 | tail -2
done; done
for v in hip f16fp8; do
  echo "== $v config 5"; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so timeout 600 python bench.py --config 5 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d.get('ms_per_step'), {k: d.get(k) for k in ('value',)}, json.dumps(d)[:600])"
done
