#!/bin/bash
# round 6, first GPU call: new composition / determinism tests, the two-workgroup schedule of dcn_bwdw6, a default bench line
mkdir -p gpurun_out
L=gpurun_out/r06_first.log
: > $L
rocminfo 2>/dev/null | grep -m1 -E "Marketing" >> $L
timeout 1500 python -m pytest tests/test_gpu_dcn.py -x -q -m gpu -k "composition or deterministic" -rA 2>&1 | grep -E "passed|failed|PASS|FAIL|DIFFERS|bit-identical|Error" | tail -40 >> $L
echo "--- determinism_check WG=2" >> $L
RVSR_BWDW6_WG=2 timeout 600 python tests/determinism_check.py >> $L 2>&1
echo "--- determinism_check WG=1" >> $L
timeout 600 python tests/determinism_check.py >> $L 2>&1
echo "--- bench" >> $L
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r06_first_bench.json 2>> $L
tail -c 1500 gpurun_out/r06_first_bench.json >> $L
cat $L
