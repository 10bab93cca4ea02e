"""Helpers shared by oracle/make_golden.py and the tests that read its fixtures.  TEST
INFRASTRUCTURE ONLY (no reference import here, so it loads on the GPU box too)."""
import torch

SLICE = 64


def sample_slice(g):
    """First SLICE values plus SLICE values at a fixed stride over the flattened tensor."""
    f = g.detach().reshape(-1)
    stride = max(1, f.numel() // SLICE)
    return torch.cat([f[:SLICE], f[::stride][:SLICE]]).numpy()
