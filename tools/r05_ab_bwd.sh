#!/bin/bash
# round 5: DCN backward generations inside the driver's step, one gpurun call, two interleaved repetitions
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for cfg in "RVSR_DCN_BWD=6 RVSR_DCN_BWDW=4" "RVSR_DCN_BWD=7 RVSR_DCN_BWDW=4" "RVSR_DCN_BWD=7 RVSR_DCN_BWDW=6"; do
    env $cfg timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sweep --no-extra "$@" > gpurun_out/ab/x.json 2> gpurun_out/ab/x.err || tail -5 gpurun_out/ab/x.err
    python - "$cfg" $rep <<'PY'
import json, sys
r=json.loads([l for l in open('gpurun_out/ab/x.json') if l.startswith('{')][-1])
print('%s rep %s: ms/step %.2f  dcn_fwd frac %.4f (%.4f ms)  dcn_bwd %.2f ms  conv %.4f ms' % (sys.argv[1], sys.argv[2], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['roofline']['dcn_bwd_ms_per_step'], r['roofline_conv']['avg_launch_ms']))
PY
  done
done
