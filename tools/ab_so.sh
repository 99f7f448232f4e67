#!/bin/bash
# A/B of two library builds inside one gpurun call: tools/ab_so.sh <A.so> <B.so> [bench flags...]   (paths relative to realvsr_amd/csrc)
A="$1"; B="$2"; shift 2
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in A B; do
    [ $v = A ] && SO="$A" || SO="$B"
    RVSR_SO=$PWD/realvsr_amd/csrc/$SO timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep "$@" > gpurun_out/ab/$v$rep.json 2> gpurun_out/ab/$v$rep.err || tail -5 gpurun_out/ab/$v$rep.err
    python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/ab/$v$rep.json') if l.startswith('{')][-1])
print('$v$rep [$SO] ms/step %.2f  dcn_fwd frac %.4f (%.4f ms)  dcn_bwd %.2f ms  conv %.4f ms frac %.4f' % (r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['roofline']['dcn_bwd_ms_per_step'], r['roofline_conv']['avg_launch_ms'], r['roofline_conv']['frac']))
PY
  done
done
