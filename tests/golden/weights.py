"""Deterministic weight fill shared by make_golden.py and the tests.

numpy's legacy ``RandomState`` stream is frozen across numpy versions, so a model whose
state_dict is too large to commit (EDVR_NoUp needs nf=64 -> 6 MB) can be re-created
bit-identically from (parameter names, shapes, seed) on any box.
"""
import numpy as np
import torch


def fill_state_dict(module, seed, offset_std=0.02):
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        vals = rs.standard_normal(tuple(t.shape)).astype(np.float32)
        if name.endswith('bias'):
            vals *= 0.01
        elif 'conv_offset_mask' in name:
            vals *= offset_std
        else:
            fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1
            vals *= 0.5 / np.sqrt(fan_in)
        sd[name] = torch.from_numpy(vals)
    module.load_state_dict(sd)
    return module
