#!/usr/bin/env python
"""Copy the artefacts of tools/profiles.sh (gpurun_out/<round>_profiles/) into profiles/ under <round>_ names and derive
  <round>_dcn_fwd_pmc.json / <round>_dcn_fwd_pmc_nf128.json  the DCN-forward HBM-traffic records bench.py rescales (FETCH_SIZE x2 + WRITE_SIZE,
                                                      MI355X_MICROARCH.md HBM section), offset std 1.25 px;
  <round>_dcn_sq_counters.json                            SQ counters of dcn_fwd3 at nf64 / nf128;
  <round>_dcn_bwd_sq_counters.json                        SQ counters + HBM traffic of the backward pair dcn_bwdin6 / dcn_bwdw6.
A kernel family launches several template instantiations per call (halo candidates that return at once unless selected): the record keeps,
per counter, the instantiation that did the work (the largest mean)."""
import glob
import json
import os
import re
import shutil

import sys
R = sys.argv[1] if len(sys.argv) > 1 else 'r06'
SRC, DST = 'gpurun_out/%s_profiles' % R, 'profiles'
for name in ['bench_default.json', 'bench_c3.json', 'infer_c5.json', 'bench_force_allreduce.json', 'default_kernel_stats.csv', 'c3_kernel_stats.csv']:
    src = os.path.join(SRC, name)
    if not os.path.exists(src):
        continue
    if name.endswith('.json'):      # keep only the JSON line
        lines = [l for l in open(src) if l.startswith('{')]
        if lines:
            with open(os.path.join(DST, R + '_' + name), 'w') as f:
                f.write(lines[-1])
    else:
        shutil.copy(src, os.path.join(DST, R + '_' + name))


def counters(tag, family):
    c = {}
    for f in sorted(glob.glob(os.path.join(SRC, 'pmc_%s_p*.txt' % tag))):
        for line in open(f):
            m = re.match(r'^(.*?)\s+([A-Z_0-9]+)\s+(\d+)\s+\(mean of (\d+)', line)
            if m and family in m.group(1):
                c[m.group(2)] = max(c.get(m.group(2), 0), int(m.group(3)))
    return c


def describe(k, c):
    if 'SQ_WAVE_CYCLES' not in c or 'SQ_WAIT_ANY' not in c:
        return
    wc = 4.0 * c['SQ_WAVE_CYCLES']
    print(k, 'instructions per wave: VALU %d (of them MFMA %d) SALU %d LDS %d VMEM %d' % (
        c['SQ_INSTS_VALU'] / c['SQ_WAVES'], c['SQ_INSTS_MFMA'] / c['SQ_WAVES'], c['SQ_INSTS_SALU'] / c['SQ_WAVES'], c['SQ_INSTS_LDS'] / c['SQ_WAVES'],
        (c['SQ_INSTS_VMEM_RD'] + c['SQ_INSTS_VMEM_WR']) / c['SQ_WAVES']),
        '| of the wave cycles: issuing %.2f, waiting at s_waitcnt / barrier %.2f, issue-stalled %.2f' % (
            4.0 * c['SQ_ACTIVE_INST_ANY'] / wc, 4.0 * c['SQ_WAIT_ANY'] / wc, 4.0 * c['SQ_WAIT_INST_ANY'] / wc),
        '| LDS busy cycles per CU %.0f M, bank conflicts %.2f of them' % (c['SQ_LDS_IDX_ACTIVE'] / 256e6, c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']))


NOTE = ('mean per dispatch, rocprofv3 --pmc passes (five separate runs per shape: three SQ groups, FETCH_SIZE, WRITE_SIZE) of tools/dcn_micro.py '
        '--iters 2 --ostd 1.25 at the L1 shape; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, '
        'FETCH/WRITE_SIZE KB (FETCH_SIZE x2 for bytes on gfx950; it counts fabric requests, Infinity-Cache hits included); tools/profiles.sh')
std = 1.25
fwd_summary = {}
for tag, B, C, out in (('fwd64', 40, 64, R + '_dcn_fwd_pmc.json'), ('fwd128', 16, 128, R + '_dcn_fwd_pmc_nf128.json')):
    fwd = counters(tag, 'dcn_fwd3')
    fwd_summary['dcn_fwd3_kernel nf%d B=%d' % (C, B)] = fwd
    describe('dcn_fwd3 nf%d' % C, fwd)
    if 'FETCH_SIZE' in fwd and 'WRITE_SIZE' in fwd:
        px = B * 180 * 320
        alg = 4 * (C + 216 + C)
        hbm = (2 * fwd['FETCH_SIZE'] + fwd['WRITE_SIZE']) * 1024
        rec = {'kernel': 'dcn_fwd3_kernel', 'shape': {'B': B, 'C': C, 'Co': C, 'dg': 8, 'H': 180, 'W': 320, 'offset_std_px': std},
               'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/dcn_micro.py --iters 2 --B %d --C %d --ostd %s --fwd-only '
                          '(and a separate --pmc WRITE_SIZE pass); tools/profiles.sh' % (B, C, std),
               'FETCH_SIZE_KB_reported': fwd['FETCH_SIZE'], 'fetch_correction': 'x2 (gfx950 FETCH_SIZE reports 1/2 of coalesced reads; calibrated in round 1, profiles/r01_notes.md)',
               'WRITE_SIZE_KB_reported': fwd['WRITE_SIZE'], 'hbm_bytes_per_launch': hbm, 'pixels_per_launch': px,
               'hbm_bytes_per_pixel': round(hbm / px, 1), 'algorithmic_bytes_per_pixel': alg}
        with open(os.path.join(DST, out), 'w') as f:
            json.dump(rec, f, indent=2)
        print(out, 'HBM bytes/px', rec['hbm_bytes_per_pixel'], 'vs algorithmic', alg)
with open(os.path.join(DST, R + '_dcn_sq_counters.json'), 'w') as f:
    json.dump({'note': NOTE + ' --fwd-only', 'counters': fwd_summary}, f, indent=1)

bwd_summary = {}
for tag, B, C in (('bwd64', 40, 64), ('bwd128', 16, 128)):
    for fam in ('dcn_bwdin6', 'dcn_bwdw6'):
        c = counters(tag, fam)
        if c:
            px = B * 180 * 320
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                c['hbm_bytes_per_pixel'] = round((2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / px, 1)
            bwd_summary['%s_kernel nf%d B=%d' % (fam, C, B)] = c
            describe('%s nf%d' % (fam, C), c)
with open(os.path.join(DST, R + '_dcn_bwd_sq_counters.json'), 'w') as f:
    json.dump({'note': NOTE + ' (forward + backward of the pack; algorithmic bytes per pixel of the whole backward: 4 (2 C + 432 + Co) = %d at nf64, '
                              '%d at nf128; dcn_bwdin6 additionally writes and dcn_bwdw6 reads the 4 Co B/px transposed-gradient hand-off)' % (4 * (128 + 432 + 64), 4 * (256 + 432 + 128)),
               'counters': bwd_summary}, f, indent=1)
