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


_checked_kernels = {}


def _kernel_gain(kernel):
    """The HIP kernels hold the binomial weights as constants; a ``kernel`` argument (reference signature) must be
    gain * gauss_kernel().  The values are read back ONCE per tensor (id, version): reference-style callers pass a
    fresh gauss_kernel(device=cuda) per loss call and a host sync per call would stall the stream."""
    if kernel is None:
        return 1.0
    if kernel.dim() != 4 or kernel.shape[-2:] != (5, 5):
        raise NotImplementedError('only the 5x5 binomial gauss_kernel() is implemented')
    key = (kernel.data_ptr(), kernel._version, kernel.shape[0])
    gain = _checked_kernels.get(key)
    if gain is None:
        k = kernel.detach().float().cpu()
        gain = float(k[0, 0, 2, 2]) * 256. / 36.
        ref = gauss_kernel(channels=k.shape[0]) * gain
        if k.shape[1] != 1 or gain == 0. or (k - ref).abs().max() > 1e-6 * abs(gain):
            raise NotImplementedError('only (a multiple of) the 5x5 binomial gauss_kernel() is implemented')
        if len(_checked_kernels) > 64:
            _checked_kernels.clear()
        _checked_kernels[key] = gain
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
