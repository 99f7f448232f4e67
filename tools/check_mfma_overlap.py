#!/usr/bin/env python
"""Scan the gfx950 ISA of the library's kernels for matrix-core instructions whose destination registers overlap a source operand
(A or B) without being the tied accumulator: `v_mfma ... v[98:113], v[110:113], v[150:153], 0`.  hipcc (ROCm 7.2) produces that when the
first MFMA of an accumulator takes SrcC = 0 and an operand dies at the instruction; on the MI355X the products of the upper operand half
then come out wrong from run to run (profiles/r05_notes.md).   usage: tools/check_mfma_overlap.py [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'realvsr_amd', 'csrc')
files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(root, '*.hip')))
rng = lambda s: (lambda m: (int(m.group(2)), int(m.group(3) or m.group(2))))(re.match(r'([va])\[?(\d+):?(\d+)?\]?', s))
bad = 0
for f in files:
    with tempfile.NamedTemporaryFile(suffix='.s') as t:
        subprocess.run(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-munsafe-fp-atomics', '-Wno-unused-function', '-S',
                        '--cuda-device-only', f, '-o', t.name], check=True, stderr=subprocess.DEVNULL, cwd=root)
        kern = None
        for line in open(t.name):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                kern = m.group(1)
            m = re.match(r'\s+(v_mfma\w+)\s+([va]\[[\d:]+\]),\s*([va]\[?[\d:]+\]?),\s*([va]\[?[\d:]+\]?),\s*(\S+)', line)
            if not m:
                continue
            d, a, b, c = m.group(2), m.group(3), m.group(4), m.group(5)
            if c == d:
                continue
            (d0, d1), kind = rng(d), d[0]
            for name, s in (('A', a), ('B', b)):
                if s[0] == kind:
                    s0, s1 = rng(s)
                    if s0 <= d1 and d0 <= s1:
                        bad += 1
                        print('%s: %s: %s' % (os.path.basename(f), (kern or '?')[:70], line.strip()))
print('%d overlapping matrix-core instruction(s)' % bad)
sys.exit(1 if bad else 0)
