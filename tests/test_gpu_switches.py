"""The developer switches that select the previous kernel generation or a fixed window / tile halo (realvsr_amd/csrc: RVSR_DCN_BWD,
RVSR_BWDW6_WG, RVSR_DCN3_HALO, RVSR_DCN5_HALO; realvsr_amd: RVSR_PACK_CACHE) still produce reference arithmetic: the previous generation is
also the fallback for calls the newest kernels do not take, and every window size is a kernel of its own that the device-side selection only
reaches at the matching offset scale.  The switches are read once per process, so each setting runs tests/switch_check.py in a
subprocess.  (Round 5 moved the older generations out of the library: experiments/.)  -m gpu"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SETTINGS = ['RVSR_DCN_BWD=6',        # dcn_bwdin5 + dcn_bwdw4 (round 4's pair; dcn_bwdw6 needs dcn_bwdin6's operand buffer)
            'RVSR_DCN_BWD=64',       # dcn_bwdin6 + dcn_bwdw4
            'RVSR_BWDW6_WG=2',       # dcn_bwdw6 as two 4-wave workgroups per CU (4-row tiles, 2 px window)
            'RVSR_DCN5_HALO=2', 'RVSR_DCN5_HALO=4', 'RVSR_DCN5_HALO=5', 'RVSR_DCN5_HALO=8', 'RVSR_DCN5_HALO=12',          # dcn_bwdin6's windows
            'RVSR_DCN_BWD=6,RVSR_DCN5_HALO=2', 'RVSR_DCN_BWD=6,RVSR_DCN5_HALO=5', 'RVSR_DCN_BWD=6,RVSR_DCN5_HALO=12',   # dcn_bwdin5's
            'RVSR_DCN3_HALO=3', 'RVSR_DCN3_HALO=7', 'RVSR_DCN3_HALO=11', 'RVSR_PACK_CACHE=0']


@pytest.mark.parametrize('setting', SETTINGS)
def test_switch_keeps_parity(setting):
    env = dict(os.environ, **dict(kv.split('=') for kv in setting.split(',')))
    out = subprocess.run([sys.executable, os.path.join(HERE, 'switch_check.py')], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-400:])
    assert out.returncode == 0, (setting, out.stdout[-800:], out.stderr[-1500:])
