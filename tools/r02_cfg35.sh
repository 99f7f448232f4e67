#!/bin/bash
# config 3 / config 5 lines + nf128 DCN counters
mkdir -p gpurun_out/r02c
FILES="fullsize infer" X="" T=900 bash tools/gpu_tests.sh 2>&1 | grep -E "exit|passed|failed|FAILED|Error" 
timeout 900 python bench.py --nf 128 --nframes 7 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02c/bench_c3.json 2> gpurun_out/r02c/bench_c3.err; tail -c 1800 gpurun_out/r02c/bench_c3.json; tail -3 gpurun_out/r02c/bench_c3.err
timeout 900 python bench.py --nf 128 --nframes 7 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --offset-px 1 > gpurun_out/r02c/bench_c3_1px.json 2>/dev/null; tail -c 900 gpurun_out/r02c/bench_c3_1px.json
timeout 900 python tools/infer_clip.py > gpurun_out/r02c/infer_c5.json 2> gpurun_out/r02c/infer_c5.err; cat gpurun_out/r02c/infer_c5.json; tail -3 gpurun_out/r02c/infer_c5.err
