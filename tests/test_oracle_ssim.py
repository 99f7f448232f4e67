"""oracle/ssim_oracle.py (restated third-party SSIM, PARITY UNPINNED): identities + float64 finite differences.  CPU."""
import torch

from oracle import ssim_oracle as S
from oracle import edvr_oracle as O


def test_window_is_the_11x11_sigma_1p5_gaussian():
    w = S.fspecial_gauss(11, 1.5, 2)
    assert w.shape == (2, 1, 11, 11) and w.dtype == torch.float32
    assert abs(float(w[0].sum()) - 1.0) < 1e-6
    assert torch.equal(w[0], w[1]) and torch.equal(w[0, 0], w[0, 0].t())
    assert abs(float(w[0, 0, 5, 5] / w[0, 0, 5, 4]) - float(torch.exp(torch.tensor(1 / 4.5)))) < 1e-5


def test_identities():
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(2, 1, 24, 31, generator=g), torch.rand(2, 1, 24, 31, generator=g)
    assert abs(float(S.ssim_loss(x, x))) < 1e-6                     # SSIM(x, x) = 1
    assert abs(float(S.ssim_loss(x, y) - S.ssim_loss(y, x))) < 1e-7  # symmetric
    l = float(S.ssim_loss(x, y))
    assert 0.0 < l <= 1.0 + 1e-6                                    # cs clamped at 0 => ssim in [0, 1]
    assert float(S.ssim_loss(x, 1 - x)) > 0.9                       # anti-correlated structure: relu clamps cs to 0
    # module wrapper with the package's call signature
    m = S.SSIM(channels=1)
    assert torch.allclose(m(x, y), S.ssim_loss(x, y)) and m(x, y, as_loss=False).shape == (2,)


def test_gradient_matches_finite_differences_f64():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(1, 1, 13, 14, generator=g, dtype=torch.float64).requires_grad_(True)
    y = torch.rand(1, 1, 13, 14, generator=g, dtype=torch.float64)
    assert torch.autograd.gradcheck(lambda t: S.ssim_loss(t, y), (x,), eps=1e-6, atol=1e-6)


def test_lap_pyr_loss_ssim_mode_composition():
    g = torch.Generator().manual_seed(2)
    x, y = torch.rand(1, 1, 64, 96, generator=g), torch.rand(1, 1, 64, 96, generator=g)
    px, py = O.laplacian_pyramid(x, 3), O.laplacian_pyramid(y, 3)
    want = S.ssim_loss(px[-1], py[-1]) + O.charbonnier(px[0], py[0]) + O.charbonnier(px[1], py[1])
    assert torch.allclose(O.lap_pyr_loss(x, y, 3, lf_mode='ssim'), want)
