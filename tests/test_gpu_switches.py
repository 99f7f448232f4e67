"""Every developer switch that selects an older kernel generation or a fixed variant (realvsr_amd/csrc: RVSR_DCN_FWD, RVSR_DCN_BWD, RVSR_DCN_BWDW, RVSR_DCN3_HALO, RVSR_DCN5_HALO, RVSR_DCN_MT_WIDE, RVSR_CONV_WIDE, RVSR_CONV_FWD6, RVSR_XCD_SWIZZLE; realvsr_amd: RVSR_FLAT_GRAD_ADOPT, RVSR_PACK_CACHE, RVSR_GRAD_SINKS, RVSR_FUSE_GRAD_MASK) still produces reference
arithmetic: those kernels are also the fallbacks for geometries the newest ones do not cover.  The switches are read once per
process, so each setting runs tests/switch_check.py in a subprocess.  -m gpu"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SETTINGS = ['RVSR_DCN_FWD=2', 'RVSR_DCN_BWD=2', 'RVSR_DCN_BWD=3', 'RVSR_DCN_BWD=4',
            'RVSR_DCN_BWD=5', 'RVSR_DCN5_HALO=2', 'RVSR_DCN5_HALO=5', 'RVSR_DCN5_HALO=8', 'RVSR_DCN5_HALO=12', 'RVSR_DCN_BWDW=3',
            'RVSR_DCN_BWDW=2', 'RVSR_DCN_MT_WIDE=2', 'RVSR_XCD_SWIZZLE=0', 'RVSR_FLAT_GRAD_ADOPT=0', 'RVSR_DCN3_HALO=3', 'RVSR_DCN3_HALO=7',
            'RVSR_DCN3_HALO=11', 'RVSR_CONV_WIDE=0', 'RVSR_PACK_CACHE=0', 'RVSR_CONV_FWD6=1', 'RVSR_FUSE_GRAD_MASK=0',
            # the fourth-generation DCN forward (dcn4_kernels.hip: persistent, software-pipelined; measured, not the default -- see
            # profiles/r04_notes.md), in its workgroup shapes; the 4 px / 7 px cases of switch_check.py run its fix-up pass
            'RVSR_DCN_FWD=4',
            # conv_wgrad2 with the X rows of vertical neighbour tiles kept in LDS (a ring of six row slots): measured, no gain, off by default
            'RVSR_WGRAD_RING=1', 'RVSR_DCN5_HALO=4', 'RVSR_DCN5_R4=0']


@pytest.mark.parametrize('setting', SETTINGS)
def test_switch_keeps_parity(setting):
    env = dict(os.environ, **dict(kv.split('=') for kv in setting.split(',')))
    out = subprocess.run([sys.executable, os.path.join(HERE, 'switch_check.py')], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-400:])
    assert out.returncode == 0, (setting, out.stdout[-800:], out.stderr[-1500:])
