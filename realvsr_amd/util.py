"""Image-decomposition helpers with the reference's names (codes/utils/util.py:491-554).

``kernel`` arguments are accepted for signature compatibility; the binomial weights are
compile-time constants of the HIP kernels (the reference rebuilds and uploads the tensor on
every loss call).  Passing a kernel other than gauss_kernel() raises."""
import torch

from . import functional as RF


def gauss_kernel(size=5, device=torch.device('cpu'), channels=3):
    k1 = torch.tensor([1., 4., 6., 4., 1.])
    kernel = (torch.outer(k1, k1) / 256.).repeat(channels, 1, 1, 1)
    return kernel.to(device)


def _check_kernel(kernel):
    if kernel is not None and (kernel.shape[-2:] != (5, 5) or abs(float(kernel[0, 0, 2, 2]) - 36. / 256.) > 1e-7):
        raise NotImplementedError('only the 5x5 binomial gauss_kernel() is implemented')


def conv_gauss_down(img):
    return RF.pyr_down(img)


def downsample(x):
    return x[:, :, ::2, ::2]


def lap_pyramid(img, kernel=None, max_levels=3):
    _check_kernel(kernel)
    current, pyr = img, []
    for _ in range(max_levels):
        down = RF.pyr_down(current)
        pyr.append(RF.pyr_updiff(current, down))
        current = down
    return pyr


def gau_pyramid(img, kernel=None, max_levels=3):
    _check_kernel(kernel)
    current, pyr = img, [img]
    for _ in range(max_levels - 1):
        current = RF.pyr_down(current)
        pyr.append(current)
    return pyr


def laplacian_pyramid(img, kernel=None, max_levels=3):
    assert max_levels > 1
    _check_kernel(kernel)
    current, pyr = img, []
    for _ in range(max_levels - 1):
        down = RF.pyr_down(current)
        pyr.append(RF.pyr_updiff(current, down))
        current = down
    pyr.append(current)
    return pyr
