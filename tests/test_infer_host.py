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


def _apply_plan_numpy(a, b, plan):
    """What rvsr_augment_clips computes, in numpy (box paste + channel permutation; no blend)."""
    a, b = a[..., list(plan.perm), :, :], b[..., list(plan.perm), :, :]
    y0, y1, x0, x1 = plan.box
    o2 = b.copy()
    if plan.box_mode == 1:
        o2[..., y0:y1, x0:x1] = a[..., y0:y1, x0:x1]
    elif plan.box_mode == 2:
        o2 = a.copy()
        o2[..., y0:y1, x0:x1] = b[..., y0:y1, x0:x1]
    return a, o2


def test_augment_plans_match_reference_under_seeds():
    """draw_plan consumes numpy's RNG in the call order of data/augments_video_allpair.py,
    so a seeded run picks the reference's boxes / permutations (the fixture holds the reference's outputs; the device
    kernel that applies a plan is checked against the same fixture in tests/test_gpu_train.py)."""
    from realvsr_amd import augment
    g = load_golden('augment')
    for seed in range(6):
        np.random.seed(100 + seed)
        plan = augment.AugPlan()
        augment._draw_cutblur(plan, g['a4'].shape, 1.0, 0.7)
        o1, o2 = _apply_plan_numpy(g['a4'], g['b4'], plan)
        assert plan.fired and np.array_equal(o1, g['cutblur%d.1' % seed]) and np.array_equal(o2, g['cutblur%d.2' % seed])
    for seed in range(3):
        np.random.seed(200 + seed)
        plan = augment.AugPlan()
        augment._draw_rgb(plan, 1.0)
        o1, o2 = _apply_plan_numpy(g['a5'], g['b5'], plan)
        assert np.array_equal(o1, g['rgb%d.1' % seed]) and np.array_equal(o2, g['rgb%d.2' % seed])
    for seed in range(6):
        np.random.seed(300 + seed)
        plan = augment.draw_plan(g['a5'].shape, ['none', 'cutblur', 'rgb'], [1.0, 1.0, 1.0], [1.0, 0.7, 1.0], mix_p=[0.2, 0.5, 0.3])
        o1, o2 = _apply_plan_numpy(g['a5'], g['b5'], plan)
        assert np.array_equal(o1, g['mix%d.1' % seed]) and np.array_equal(o2, g['mix%d.2' % seed])
    with pytest.raises(ValueError):
        augment.cutblur(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 4))
    with pytest.raises(ValueError):
        augment.draw_plan((2, 3, 3, 12, 16), ['mixup'], [1.0], [1.0])
    np.random.seed(5)
    assert not augment.draw_plan((2, 3, 3, 12, 16), ['cutblur'], [0.0], [0.7]).fired    # prob 0 never fires
    with pytest.raises(NotImplementedError):
        augment.rgb(torch.zeros(1, 1, 3, 8, 8), torch.zeros(1, 1, 3, 8, 8))            # device op: CPU tensors refused


def test_edvr_predeblur_hr_in_schema_matches_reference():
    from realvsr_amd.archs.EDVR_arch import EDVR
    g = load_golden('edvr_predeblur')
    for tag, kw in (('pre', dict(predeblur=True, HR_in=False)), ('hr', dict(predeblur=False, HR_in=True)),
                    ('prehr', dict(predeblur=True, HR_in=True))):
        net = EDVR(nf=64, nc=3, nframes=3, groups=8, front_RBs=1, back_RBs=1, w_TSA=False, **kw)
        assert sorted(net.state_dict().keys()) == [str(k) for k in g[tag + '.keys']], tag
