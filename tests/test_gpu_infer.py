"""Test-time driver on the GPU: colour conversion bit-exact with the reference
chain, flip x4 ensemble vs the reference-generated fixture, sliding-window feature reuse bit-identical to the
reference's window-by-window loop.  -m gpu"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_util import check, dev, gemm_modes
from weights import fill_state_dict

gemm_mode = gemm_modes()
pytestmark = pytest.mark.gpu
TOL = {'f32': 5e-5, 'bf16x3': 1e-3}


def test_ycbcr_to_bgr_u8_bit_exact():
    from realvsr_amd.infer import ycbcr_to_bgr_u8
    from oracle.infer_oracle import ycbcr_to_bgr_u8 as oracle
    g = load_golden('infer')
    got = ycbcr_to_bgr_u8(torch.from_numpy(g['ycc']).to(dev())).cpu().numpy()
    assert np.array_equal(got, g['bgr_u8'])
    rng = np.random.RandomState(0)
    big = (rng.rand(3, 270, 481).astype(np.float32) * 1.2 - 0.1)      # odd sizes, out-of-range values
    assert np.array_equal(ycbcr_to_bgr_u8(torch.from_numpy(big).to(dev())[None]).cpu().numpy(), oracle(big))


def _tiny_net():
    from realvsr_amd.archs.EDVR_arch import EDVR_NoUp
    net = EDVR_NoUp(nf=64, nc=3, nframes=3, groups=8, front_RBs=1, back_RBs=1, w_TSA=True)
    fill_state_dict(net, 123)
    return net.to(dev()).eval()


def test_flip_ensemble_matches_reference(gemm_mode):
    from realvsr_amd.infer import single_forward, flipx4_forward
    g = load_golden('infer')
    net = _tiny_net()
    x = torch.from_numpy(g['flip_x']).to(dev())
    check('single_forward', single_forward(net, x), torch.from_numpy(g['flip_single']), TOL[gemm_mode])
    check('flipx4_forward', flipx4_forward(net, x), torch.from_numpy(g['flip_x4']), TOL[gemm_mode])


@pytest.mark.parametrize('padding', ['replicate', 'reflection', 'new_info', 'circle'])
def test_sliding_window_reuse_is_bit_identical(padding):
    from realvsr_amd.infer import SlidingWindowRunner
    net = _tiny_net()
    torch.manual_seed(7)
    clip = torch.rand(7, 3, 24, 40, device=dev())
    run = SlidingWindowRunner(net, 3, padding=padding, chunk=4)
    a, b = run(clip), run.reference_order(clip)
    assert a.shape == (7, 3, 24, 40)
    assert torch.equal(a, b)


def test_sliding_window_x4_and_flip():
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd.infer import SlidingWindowRunner
    net = EDVR(nf=16, nc=3, nframes=5, groups=2, front_RBs=1, back_RBs=1, w_TSA=False)
    fill_state_dict(net, 5)
    net = net.to(dev()).eval()
    torch.manual_seed(8)
    clip = torch.rand(6, 3, 16, 24, device=dev())
    run = SlidingWindowRunner(net, 5, padding='reflection', chunk=3, flip_ensemble=True)
    a, b = run(clip), run.reference_order(clip)
    assert a.shape == (6, 3, 64, 96)
    assert torch.equal(a, b)


def test_sliding_window_hipgraph_is_bit_identical():
    from realvsr_amd.infer import SlidingWindowRunner
    net = _tiny_net()
    torch.manual_seed(9)
    clip = torch.rand(6, 3, 24, 40, device=dev())
    eager = SlidingWindowRunner(net, 3, padding='reflection', chunk=4)(clip)
    run = SlidingWindowRunner(net, 3, padding='reflection', chunk=4, use_graph=True)
    assert torch.equal(run(clip), eager)
    assert torch.equal(run(clip.flip(0)), SlidingWindowRunner(net, 3, padding='reflection', chunk=4)(clip.flip(0)))  # graph reused
