"""Host logic of the test-time driver against the reference-generated fixture
tests/golden/infer.npz; runs without a GPU."""
import numpy as np
import pytest
import torch

from conftest import load_golden

MODES = ['replicate', 'reflection', 'new_info', 'circle']


def test_index_generation_matches_reference_table():
    from realvsr_amd.infer import index_generation
    table = load_golden('infer')['index_table']
    assert len(table) > 500
    for row in table:
        mi, max_n, N, crt = (int(v) for v in row[:4])
        assert index_generation(crt, max_n, N, padding=MODES[mi]) == [int(v) for v in row[4:4 + N]], row
    # the docstring example of codes/data/util.py:175-180
    assert index_generation(0, 10, 5, 'replicate') == [0, 0, 0, 1, 2]
    assert index_generation(0, 10, 5, 'reflection') == [2, 1, 0, 1, 2]
    assert index_generation(0, 10, 5, 'new_info') == [4, 3, 0, 1, 2]
    assert index_generation(0, 10, 5, 'circle') == [3, 4, 0, 1, 2]
    with pytest.raises(ValueError):
        index_generation(0, 10, 5, 'mirror')


def test_colour_oracle_matches_reference():
    from oracle.infer_oracle import ycbcr_to_bgr_u8
    g = load_golden('infer')
    got = ycbcr_to_bgr_u8(g['ycc'])
    assert got.dtype == np.uint8 and got.shape == g['bgr_u8'].shape
    assert np.array_equal(got, g['bgr_u8'])
    # video-range black / white map to 0 / 255
    assert (got[0, 0] == 0).all() and (got[2, 0] == 255).all()


def test_driver_refuses_cpu_and_bad_shapes():
    from realvsr_amd import infer
    with pytest.raises((NotImplementedError, RuntimeError)):
        infer.ycbcr_to_bgr_u8(torch.zeros(3, 8, 8))
    with pytest.raises(RuntimeError):
        infer.ycbcr_to_bgr_u8(torch.zeros(2, 8, 8))

    class _Net(object):
        center = 2
    with pytest.raises(RuntimeError):
        infer.SlidingWindowRunner(_Net(), 3)
    r = infer.SlidingWindowRunner(_Net(), 5)
    with pytest.raises(RuntimeError):
        r(torch.zeros(4, 3, 10, 16))      # H not divisible by 4


def test_tdan_state_dict_schema_matches_reference():
    from realvsr_amd.archs.TDAN_arch import TDAN
    g = load_golden('tdan')
    for tag, scale in (('s1', 1), ('s2', 2)):
        net = TDAN(channel=3, nframes=3, scale=scale, nf=64, nb_f=1, nb_b=1, groups=8)
        assert sorted(net.state_dict().keys()) == [str(k) for k in g[tag + '.keys']]
