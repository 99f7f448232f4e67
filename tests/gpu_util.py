"""Helpers shared by the -m gpu parity tests."""
import torch

from conftest import rel_err


def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def check(name, got, ref, tol):
    got = got.detach().cpu() if torch.is_tensor(got) else torch.as_tensor(got)
    ref = ref.detach().cpu() if torch.is_tensor(ref) else torch.as_tensor(ref)
    assert tuple(got.shape) == tuple(ref.shape), '%s: shape %s vs %s' % (name, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), '%s: non-finite values' % name
    e = rel_err(got, ref)
    print('%-44s rel_err %.3e (tol %.1e)' % (name, e, tol))
    assert e <= tol, '%s: rel_err %.3e > %.1e' % (name, e, tol)
    return e
