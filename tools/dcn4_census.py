#!/usr/bin/env python3
"""Instruction census of dcn_fwd4's pipeline body (nine k-steps) from the compiler's assembly listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only realvsr_amd/csrc/dcn4_kernels.hip -o /tmp/f4.s
  python tools/dcn4_census.py /tmp/f4.s ILi8ELi5ELi7ELi2ELi0E
"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
        return 'lane'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith(('s_waitcnt', 's_nop', 's_barrier')):
        return 'wait/nop'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')):
        return 'vmem'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z15dcn_fwd4_kernel' + key) and ': ' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.end_amdhsa_kernel') or lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    # the pipeline body: from the depth-1 loop header to the instruction after the 9 x 3 MT MFMAs
    hdr = [i for i, l in enumerate(body) if 'Loop Header: Depth=1' in l]
    best = None
    for h in hdr:
        n = 0
        for i in range(h, len(body)):
            if 'v_mfma' in body[i]:
                n += 1
            if i > h and 'Loop Header' in body[i]:
                break
        if best is None or n > best[1]:
            best = (h, n, i)
    h, n, e = best
    ops = collections.Counter()
    cls = collections.Counter()
    for l in body[h:e]:
        m = re.match(r'\t([a-z_0-9]+)', l)
        if not m:
            continue
        ops[m.group(1)] += 1
        cls[classify(m.group(1))] += 1
    tot = sum(cls.values())
    print(f'lines {start + h}..{start + e}: {tot} instructions, {n} MFMA; per k-step: {tot / 9:.1f}')
    print({k: round(v / 9, 1) for k, v in cls.most_common()})
    print(ops.most_common(40))


main()
