#!/usr/bin/env python
"""Copy the artefacts of tools/r03_profiles.sh (gpurun_out/r03_profiles/) into profiles/ under r03_ names and derive
the DCN-forward HBM-traffic records bench.py rescales (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section)."""
import glob
import json
import os
import re
import shutil

SRC, DST = 'gpurun_out/r03_profiles', 'profiles'
for name in ['bench_default.json', 'offsets_1px.json', 'offsets_2px.json', 'offsets_3px.json', 'offsets_5px.json', 'bench_c3.json', 'bench_c3_3px.json', 'infer_c5.json',
             'default_kernel_stats.csv', '3px_kernel_stats.csv', '5px_kernel_stats.csv', 'c3_kernel_stats.csv', 'conv_sq_counters.txt']:
    src = os.path.join(SRC, name)
    if not os.path.exists(src):
        continue
    if name.endswith('.json'):      # keep only the JSON line
        lines = [l for l in open(src) if l.startswith('{')]
        if lines:
            with open(os.path.join(DST, 'r03_' + name), 'w') as f:
                f.write(lines[-1])
    else:
        shutil.copy(src, os.path.join(DST, 'r03_' + name))


def counters(prefix):
    out = {}
    for f in sorted(glob.glob(os.path.join(SRC, prefix + '_p*.txt'))):
        for line in open(f):
            m = re.match(r'^(.*?)\s+([A-Z_0-9]+)\s+(\d+)\s+\(mean of (\d+)', line)
            if m:
                kern = m.group(1).strip().replace('void ', '')
                out.setdefault(kern, {})[m.group(2)] = int(m.group(3))
    return out


summary = {}
for prefix, C, B, std in [('pmc_C64_std0.125', 64, 40, 0.125), ('pmc_C64_std1.25', 64, 40, 1.25), ('pmc_C64_std3.75', 64, 40, 3.75),
                          ('pmc_C64_std6.25', 64, 40, 6.25), ('pmc_C128_std0.125', 128, 16, 0.125), ('pmc_C128_std3.75', 128, 16, 3.75)]:
    c = counters(prefix)
    if not c:
        continue
    summary['C%d_B%d_offset_std_%s' % (C, B, std)] = c
    fwd = next((v for k, v in c.items() if 'dcn_fwd3' in k), None)
    if fwd and 'FETCH_SIZE' in fwd and 'WRITE_SIZE' in fwd and std == 0.125:
        px = B * 180 * 320
        alg = 4 * (C + 216 + C)
        hbm = (2 * fwd['FETCH_SIZE'] + fwd['WRITE_SIZE']) * 1024
        rec = {'kernel': 'dcn_fwd3_kernel', 'shape': {'B': B, 'C': C, 'Co': C, 'dg': 8, 'H': 180, 'W': 320, 'offset_std_px': std},
               'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/dcn_micro.py --iters 2 --B %d --C %d --ostd %s '
                          '(and a separate --pmc WRITE_SIZE pass); tools/r03_profiles.sh' % (B, C, std),
               'FETCH_SIZE_KB_reported': fwd['FETCH_SIZE'], 'fetch_correction': 'x2 (gfx950 FETCH_SIZE reports 1/2 of coalesced reads; calibrated in round 1, profiles/r01_notes.md)',
               'WRITE_SIZE_KB_reported': fwd['WRITE_SIZE'], 'hbm_bytes_per_launch': hbm, 'pixels_per_launch': px,
               'hbm_bytes_per_pixel': round(hbm / px, 1), 'algorithmic_bytes_per_pixel': alg}
        with open(os.path.join(DST, 'r03_dcn_fwd_pmc%s.json' % ('' if C == 64 else '_nf128')), 'w') as f:
            json.dump(rec, f, indent=2)
        print('C', C, 'HBM bytes/px', rec['hbm_bytes_per_pixel'], 'vs algorithmic', alg)
with open(os.path.join(DST, 'r03_dcn_sq_counters.json'), 'w') as f:
    json.dump({'note': 'mean per dispatch, rocprofv3 --pmc passes of tools/dcn_micro.py (fwd + bwd of one fused DCN pack at the L1 shape); '
                       'SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, FETCH/WRITE_SIZE KB',
               'counters': summary}, f, indent=1)
print('profiles/:', sorted(os.listdir(DST)))
