#!/bin/bash
# round-2 first GPU call: new tests + offset-scaled bench baselines + rocprof stats
mkdir -p gpurun_out/r02a
FILES="train dist net" X="" bash tools/gpu_tests.sh 2>&1 | tail -60
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for px in none 1 5; do
  flag=""; [ "$px" != none ] && flag="--offset-px $px"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $flag > gpurun_out/r02a/bench_px$px.json 2> gpurun_out/r02a/bench_px$px.err
  tail -c 1500 gpurun_out/r02a/bench_px$px.json
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lf-mode cb > gpurun_out/r02a/bench_cb.json 2>&1; tail -c 600 gpurun_out/r02a/bench_cb.json
for px in 1 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/prof_px$px -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --offset-px $px > gpurun_out/r02a/prof_px$px.log 2>&1
  f=$(find gpurun_out/r02a/prof_px$px -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r02a/px${px}_kernel_stats.csv; head -12 "$f"
  find gpurun_out/r02a/prof_px$px -name '*.csv' ! -name '*kernel_stats.csv' -delete; find gpurun_out/r02a/prof_px$px -name '*.db' -delete
done
