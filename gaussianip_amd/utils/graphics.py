"""Projection helpers.  Reference: gaussiansplatting/utils/graphics_utils.py:17-20 (BasicPointCloud),
:73-93 (getProjectionMatrix), :95-99 (fov2focal / focal2fov).  Pinned by tests/golden/projection.npz."""
import math
from typing import NamedTuple

import numpy as np
import torch


class BasicPointCloud(NamedTuple):
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


def fov2focal(fov, pixels):
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal, pixels):
    return 2.0 * math.atan(pixels / (2.0 * focal))


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """OpenGL-style frustum with z mapped to [0, 1] and w = +z (z_sign = 1); float32 like the reference, which
    fills a torch.zeros(4, 4) element by element."""
    f32 = np.float32
    tx, ty = math.tan(fovX / 2.0), math.tan(fovY / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    m = torch.zeros(4, 4)
    m[0, 0] = 2.0 * znear / (right - left)
    m[1, 1] = 2.0 * znear / (top - bottom)
    m[0, 2] = (right + left) / (right - left)
    m[1, 2] = (top + bottom) / (top - bottom)
    m[3, 2] = 1.0
    m[2, 2] = zfar / (zfar - znear)
    m[2, 3] = -(zfar * znear) / (zfar - znear)
    del f32
    return m
