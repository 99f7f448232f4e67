#!/bin/bash
# A/B of one developer switch inside one gpurun call: tools/r04_ab_env.sh VAR A B [bench flags...]; two interleaved repetitions
VAR="$1"; A="$2"; B="$3"; shift 3
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $VAR=$v timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sweep --no-extra "$@" > gpurun_out/ab/${VAR}_$v$rep.json 2> gpurun_out/ab/${VAR}_$v$rep.err || tail -5 gpurun_out/ab/${VAR}_$v$rep.err
    python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/ab/${VAR}_$v$rep.json') if l.startswith('{')][-1])
print('$VAR=$v rep $rep: ms/step %.2f  dcn_fwd frac %.4f (%.4f ms)  dcn_bwd %.2f ms  conv %.4f ms frac %.4f' % (r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['roofline']['dcn_bwd_ms_per_step'], r['roofline_conv']['avg_launch_ms'], r['roofline_conv']['frac']))
PY
  done
done
