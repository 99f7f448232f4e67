#!/bin/bash
# kernel stats of the default training step in a GEMM mode: tools/r04_mode_prof.sh bf16
m=${1:-bf16}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RVSR_GEMM=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-sweep > $O/bench.json 2>/dev/null
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
head -16 $O/kernel_stats.csv | cut -c1-150
tail -1 $O/bench.json | cut -c1-300
