#!/bin/bash
# per-kernel time of the DCN micro-benchmark (fwd + bwd, B=40 L1 shape) with the product library and each named variant build
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in hip "$@"; do
  d=gpurun_out/babl_$v; rm -rf $d
  RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python tools/dcn_micro.py --iters 3 --B 40 --ostd 0.1 > /dev/null 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo -n "$v: "; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'bwdin' in r['Name'] or 'bwdw' in r['Name'] or 'dcn_fwd' in r['Name']:
        print('%s %.3f ms;' % (r['Name'].split('(')[0][-24:], float(r['AverageNs'])/1e6), end=' ')
print()
PY
  rm -rf $d
done
