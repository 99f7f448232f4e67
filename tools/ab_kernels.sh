#!/bin/bash
# Per-kernel A/B of library builds / developer switches inside ONE gpurun call, through rocprofv3 --kernel-trace --stats of tools/dcn_micro.py:
#   tools/ab_kernels.sh [-m "<dcn_micro flags>"] [-k <kernel-name regex>] name[:so[:ENV=V,ENV=V]] ...
# `so` is a file name under realvsr_amd/csrc (default: the product library).  Prints the average duration of the matching kernels per variant.
MICRO="--B 40 --iters 6 --ostd 1.25"
KRE="dcn_bw|reduce_partials"
while getopts "m:k:" o; do case $o in m) MICRO="$OPTARG";; k) KRE="$OPTARG";; esac; done
shift $((OPTIND - 1))
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for rep in 1 2; do
for spec in "$@"; do
  IFS=: read -r name so envs <<< "$spec"
  [ -z "$so" ] && so=librealvsr_hip.so
  rm -rf gpurun_out/prof_ab
  ( export RVSR_SO=$PWD/realvsr_amd/csrc/$so
    for kv in ${envs//,/ }; do export "$kv"; done
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab --output-format csv -- python tools/dcn_micro.py $MICRO > gpurun_out/prof_ab.log 2>&1 )
  f=$(ls gpurun_out/prof_ab/*/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "$name: no stats"; tail -5 gpurun_out/prof_ab.log; continue; fi
  python - "$f" "$name" "$KRE" "$rep" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[3], r['Name'])]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) / max(1, 6 + 1) for r in rows)
print('%-14s rep%s | ' % (sys.argv[2], sys.argv[4]) + '; '.join('%s %.3f ms x%s' % (re.sub(r'^void |\(.*$', '', r['Name'])[:34], float(r['AverageNs']) / 1e6, r['Calls']) for r in rows[:5]))
PY
done
done
rm -rf gpurun_out/prof_ab
