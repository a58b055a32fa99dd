"""Shared helpers for the test-suite."""
import copy
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def rel_err(got, ref):
    """max |got - ref| / max |ref|  (the north-star's "rel" on the output scale)"""
    got = torch.as_tensor(got).detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float().cpu()
    assert got.shape == ref.shape, "shape %s vs %s" % (tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), "non-finite values"
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def assert_close(got, ref, tol, what):
    e = rel_err(got, ref)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)
    return e


def cfg_copy(cfg):
    return copy.deepcopy(cfg)
