"""Shared helpers for the tests: fixture loading and oracle model reconstruction."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def batch_from(fx, prefix="batch_"):
    return {k[len(prefix):]: torch.tensor(v) for k, v in fx.items() if k.startswith(prefix)}


def mlp_from_state(fx, prefix="sd_", root="output_scalars.mlp.model."):
    """Split the reference Sequential state into [(W,b)] linears and [(g,beta)] layer norms."""
    idx = sorted({int(k[len(prefix + root):].split(".")[0]) for k in fx if k.startswith(prefix + root)})
    lin, norm = [], []
    for i in idx:
        w = torch.tensor(fx[f"{prefix}{root}{i}.weight"])
        b = torch.tensor(fx[f"{prefix}{root}{i}.bias"])
        (lin if w.ndim == 2 else norm).append((w, b))
    return lin, norm


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
