"""Small numeric helpers.  Reference: gaussiansplatting/utils/general_utils.py — inverse_sigmoid :18,
get_expon_lr_func :29-62, strip_lowerdiag / strip_symmetric :64-76, build_rotation :78-99,
build_scaling_rotation :101-110.  Unlike the reference these follow the device of their inputs instead of
hard-coding "cuda".  Pinned by tests/golden/covariance.npz and lr_schedule.npz."""
import math

import numpy as np
import torch


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final over max_steps, optionally eased in by a sine ramp that
    starts at lr_delay_mult.  Negative steps or an all-zero schedule disable the parameter (rate 0)."""
    off = lr_init == 0.0 and lr_final == 0.0

    def rate(step):
        if off or step < 0:
            return 0.0
        ramp = 1.0
        if lr_delay_steps > 0:
            ramp = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return ramp * np.exp((1 - t) * np.log(lr_init) + t * np.log(lr_final))

    return rate


_TRIU = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))


def strip_lowerdiag(L):
    """[N,3,3] symmetric -> [N,6] in the order xx, xy, xz, yy, yz, zz (the rasterizer's cov3D packing)."""
    return torch.stack([L[:, i, j] for i, j in _TRIU], dim=1).to(torch.float32)


def strip_symmetric(sym):
    return strip_lowerdiag(sym)


def build_rotation(r):
    """Rotation matrices from quaternions (w, x, y, z); the quaternion is normalised here."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = (1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y))
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def build_scaling_rotation(s, r):
    """L = R diag(s)."""
    return build_rotation(r) * s.to(torch.float32)[:, None, :]


del math
