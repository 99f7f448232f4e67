#!/usr/bin/env python
"""Copy the artefacts of tools/r04_profiles.sh (gpurun_out/r04_profiles/) into profiles/ under r04_ names and derive the DCN-forward
HBM-traffic record bench.py rescales (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section) and the SQ counter summary of both
forward generations at 1.25 px offsets."""
import glob
import json
import os
import re
import shutil

SRC, DST = 'gpurun_out/r04_profiles', 'profiles'
for name in ['bench_default.json', 'offsets_3px.json', 'bench_c3.json', 'infer_c5.json', 'default_kernel_stats.csv', 'c3_kernel_stats.csv', 'bf16_mode_kernel_stats.csv']:
    src = os.path.join(SRC, name)
    if not os.path.exists(src):
        continue
    if name.endswith('.json'):      # keep only the JSON line
        lines = [l for l in open(src) if l.startswith('{')]
        if lines:
            with open(os.path.join(DST, 'r04_' + name), 'w') as f:
                f.write(lines[-1])
    else:
        shutil.copy(src, os.path.join(DST, 'r04_' + name))

summary = {}
for gen in (3, 4):
    c = {}
    for f in sorted(glob.glob(os.path.join(SRC, 'pmc_fwd%d_p*.txt' % gen))):
        for line in open(f):
            m = re.match(r'^(.*?)\s+([A-Z_0-9]+)\s+(\d+)\s+\(mean of (\d+)', line)
            if m and 'dcn_fwd%d' % gen in m.group(1):
                c[m.group(2)] = int(m.group(3))
    summary['dcn_fwd%d_kernel' % gen] = c
B, C, std = 40, 64, 1.25
fwd = summary['dcn_fwd3_kernel']
if 'FETCH_SIZE' in fwd and 'WRITE_SIZE' in fwd:
    px = B * 180 * 320
    alg = 4 * (C + 216 + C)
    hbm = (2 * fwd['FETCH_SIZE'] + fwd['WRITE_SIZE']) * 1024
    rec = {'kernel': 'dcn_fwd3_kernel', 'shape': {'B': B, 'C': C, 'Co': C, 'dg': 8, 'H': 180, 'W': 320, 'offset_std_px': std},
           'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/dcn_micro.py --iters 2 --B %d --C %d --ostd %s --fwd-only '
                      '(and a separate --pmc WRITE_SIZE pass); tools/r04_profiles.sh' % (B, C, std),
           'FETCH_SIZE_KB_reported': fwd['FETCH_SIZE'], 'fetch_correction': 'x2 (gfx950 FETCH_SIZE reports 1/2 of coalesced reads; calibrated in round 1, profiles/r01_notes.md)',
           'WRITE_SIZE_KB_reported': fwd['WRITE_SIZE'], 'hbm_bytes_per_launch': hbm, 'pixels_per_launch': px,
           'hbm_bytes_per_pixel': round(hbm / px, 1), 'algorithmic_bytes_per_pixel': alg}
    with open(os.path.join(DST, 'r04_dcn_fwd_pmc.json'), 'w') as f:
        json.dump(rec, f, indent=2)
    print('HBM bytes/px', rec['hbm_bytes_per_pixel'], 'vs algorithmic', alg)
with open(os.path.join(DST, 'r04_dcn_sq_counters.json'), 'w') as f:
    json.dump({'note': 'mean per dispatch, rocprofv3 --pmc passes of tools/dcn_micro.py --fwd-only --B 40 --ostd 1.25 (L1 shape) with RVSR_DCN_FWD=3 / 4; '
                       'SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, FETCH/WRITE_SIZE KB '
                       '(FETCH_SIZE x2 for bytes; it counts fabric requests, Infinity-Cache hits included)',
               'counters': summary}, f, indent=1)
for k, c in summary.items():
    if not c:
        continue
    wc = 4.0 * c['SQ_WAVE_CYCLES']
    print(k, 'instructions per wave: VALU %d (of them MFMA %d) SALU %d LDS %d VMEM %d' % (c['SQ_INSTS_VALU'] / c['SQ_WAVES'], c['SQ_INSTS_MFMA'] / c['SQ_WAVES'],
          c['SQ_INSTS_SALU'] / c['SQ_WAVES'], c['SQ_INSTS_LDS'] / c['SQ_WAVES'], (c['SQ_INSTS_VMEM_RD'] + c['SQ_INSTS_VMEM_WR']) / c['SQ_WAVES']),
          '| of the wave cycles: issuing %.2f, waiting at s_waitcnt / barrier %.2f, issue-stalled %.2f' % (4.0 * c['SQ_ACTIVE_INST_ANY'] / wc, 4.0 * c['SQ_WAIT_ANY'] / wc, 4.0 * c['SQ_WAIT_INST_ANY'] / wc),
          '| LDS busy cycles per CU %.0f M, bank conflicts %.2f of them' % (c['SQ_LDS_IDX_ACTIVE'] / 256e6, c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']))
