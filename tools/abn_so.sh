#!/bin/bash
# A/B/C... of several library builds inside one gpurun call: tools/abn_so.sh "<bench flags>" A.so B.so ...   (paths relative to realvsr_amd/csrc)
FLAGS="$1"; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for SO in "$@"; do
    RVSR_SO=$PWD/realvsr_amd/csrc/$SO timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep $FLAGS > gpurun_out/ab/x.json 2> gpurun_out/ab/x.err || tail -5 gpurun_out/ab/x.err
    python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/ab/x.json') if l.startswith('{')][-1])
print('rep$rep [$SO] ms/step %.2f  dcn_fwd frac %.4f (%.4f ms)  dcn_bwd %.2f ms  conv %.4f ms frac %.4f' % (r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['roofline']['dcn_bwd_ms_per_step'], r['roofline_conv']['avg_launch_ms'], r['roofline_conv']['frac']))
PY
  done
done
