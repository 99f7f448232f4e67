#!/bin/bash
# Run the -m gpu suite file by file (separate processes + timeouts so a faulting kernel cannot
# hide the other results); logs land in gpurun_out/.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build.log; }
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" 
for f in ${FILES:-conv dcn misc net infer fullsize train dist}; do
  timeout ${T:-600} python -m pytest tests/test_gpu_$f.py -m gpu -q -rA ${X:--x} --tb=short > gpurun_out/test_$f.log 2>&1
  echo "== test_gpu_$f exit $? =="; grep -E "rel_err|passed|failed|Error|error" gpurun_out/test_$f.log | tail -${N:-40}
done
