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
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(np.asarray(a)) else np.float64)
    b = np.asarray(b, dtype=np.complex128 if np.iscomplexobj(np.asarray(b)) else np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def philox_mask(seed: int, step: int, length: int, scale: float):
    """numpy restatement of gfdn_draw_mask (include/diffgfdn_hip.h): bit t of the mask is bit
    (t mod 128) of Philox4x32-10(counter = (t // 128, 0, step_lo, step_hi), key = seed)."""
    n = (length + 127) // 128
    M = np.uint64(0xFFFFFFFF)
    c = [np.arange(n, dtype=np.uint64), np.zeros(n, np.uint64),
         np.full(n, step & 0xFFFFFFFF, np.uint64), np.full(n, (step >> 32) & 0xFFFFFFFF, np.uint64)]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & M, p1 >> np.uint64(32), p1 & M
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    words = np.stack(c, axis=1).reshape(-1).astype(np.uint32)              # word 4 i + w
    bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[:length]
    count = int(bits.sum())
    w = np.float32(scale) / np.float32(count) if count else np.float32(0)
    return bits.astype(np.float32) * w, count
