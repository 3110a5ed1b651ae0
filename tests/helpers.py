"""Shared test helpers (golden fixture loading, tolerances)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def params_from(d, prefix="p.", dtype=torch.float32, device="cpu"):
    return {k[len(prefix):]: torch.from_numpy(v).to(device=device, dtype=dtype)
            for k, v in d.items() if k.startswith(prefix)}


def t(a, dtype=None, device="cpu"):
    x = torch.from_numpy(np.asarray(a))
    if dtype is not None and x.is_floating_point():
        x = x.to(dtype)
    return x.to(device)


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the 'rel-err' north_star quotes (tensor-level)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = max(b.abs().max().item(), 1e-30)
    return (a - b).abs().max().item() / denom


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2: robust to the handful of ReLU-mask flips that dominate a max-norm comparison of
    gradients of deep batch-normalised conv stacks."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
