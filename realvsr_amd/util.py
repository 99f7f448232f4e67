"""Image-decomposition helpers with the reference's names (codes/utils/util.py:491-554).

``kernel`` arguments are accepted for signature compatibility; the binomial weights are
compile-time constants of the HIP kernels (the reference rebuilds and uploads the tensor on
every loss call).  Passing a kernel other than gauss_kernel() raises."""
import weakref

import torch

from . import functional as RF


def gauss_kernel(size=5, device=torch.device('cpu'), channels=3):
    k1 = torch.tensor([1., 4., 6., 4., 1.])
    kernel = (torch.outer(k1, k1) / 256.).repeat(channels, 1, 1, 1).to(device)
    kernel._rvsr_gain = 1.0   # built here: _kernel_gain need not read it back (valid while _version stays 0)
    return kernel


_checked_kernels = {}   # id(tensor) -> (weakref to the tensor, version, gain)


def _kernel_gain(kernel):
    """The HIP kernels hold the binomial weights as constants; a ``kernel`` argument (reference signature) must be
    gain * gauss_kernel().  The values are read back ONCE per live tensor object and version: the cache entry holds a weak
    reference to the tensor it validated, so a different tensor that the caching allocator later places at the same
    address (reference-style callers build a fresh gauss_kernel(device=cuda) per loss call) is validated again instead
    of inheriting a stale gain; a host sync per call would stall the stream."""
    if kernel is None:
        return 1.0
    if kernel.dim() != 4 or kernel.shape[-2:] != (5, 5):
        raise NotImplementedError('only the 5x5 binomial gauss_kernel() is implemented')
    if getattr(kernel, '_rvsr_gain', None) is not None and kernel._version == 0:
        return kernel._rvsr_gain
    hit = _checked_kernels.get(id(kernel))
    if hit is not None and hit[0]() is kernel and hit[1] == kernel._version:
        return hit[2]
    k = kernel.detach().float().cpu()
    gain = float(k[0, 0, 2, 2]) * 256. / 36.
    ref = gauss_kernel(channels=k.shape[0]) * gain
    if k.shape[1] != 1 or gain == 0. or (k - ref).abs().max() > 1e-6 * abs(gain):
        raise NotImplementedError('only (a multiple of) the 5x5 binomial gauss_kernel() is implemented')
    for key in [key for key, v in _checked_kernels.items() if v[0]() is None]:
        del _checked_kernels[key]
    _checked_kernels[id(kernel)] = (weakref.ref(kernel), kernel._version, gain)
    return gain


def _check_kernel(kernel):
    if _kernel_gain(kernel) != 1.0:
        raise NotImplementedError('the pyramid builders take gauss_kernel() itself')


def conv_gauss(img, kernel=None):
    """Reflect-pad 2 + depthwise 5x5 (utils/util.py:503-506); ``kernel`` = gain * gauss_kernel()."""
    return RF.conv_gauss(img, _kernel_gain(kernel))


def conv_gauss_down(img):
    return RF.pyr_down(img)


def upsample(x):
    """Zero-insert x2 + conv_gauss with 4 * gauss_kernel (utils/util.py:513-516)."""
    return RF.pyr_upsample(x)


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
