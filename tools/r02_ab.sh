#!/bin/bash
# A/B of an env switch inside one gpurun call: tools/r02_ab.sh "<ENV_A>" "<ENV_B>" [bench flags...]
A="$1"; B="$2"; shift 2
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in A B; do
    [ $v = A ] && E="$A" || E="$B"
    env $E timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab/$v$rep.json 2> gpurun_out/ab/$v$rep.err || tail -5 gpurun_out/ab/$v$rep.err
    python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/ab/$v$rep.json') if l.startswith('{')][-1])
print('$v$rep [%s] ms/step %.2f  dcn_fwd frac %.4f avg_launch %.4f ms  off %.3f px' % ('$E', r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['offset_abs_mean_px']))
PY
  done
done
