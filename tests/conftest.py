import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: z[k] for k in z.files}


def golden_sd(g, prefix='sd.'):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
