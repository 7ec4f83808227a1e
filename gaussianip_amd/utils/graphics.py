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


def get_projection_matrix(fovy, aspect_wh, near, far):
    """threestudio's OpenGL projection with the y axis flipped (threestudio/utils/ops.py:266-278); fovy [B] radians.
    The data module builds the per-view mvp matrices from it with near 0.1 / far 1000 (camera_data.py:462-463)."""
    B = fovy.shape[0]
    proj = torch.zeros(B, 4, 4, dtype=torch.float32)
    proj[:, 0, 0] = 1.0 / (torch.tan(fovy / 2.0) * aspect_wh)
    proj[:, 1, 1] = -1.0 / torch.tan(fovy / 2.0)
    proj[:, 2, 2] = -(far + near) / (far - near)
    proj[:, 2, 3] = -2.0 * far * near / (far - near)
    proj[:, 3, 2] = -1.0
    return proj


def get_mvp_matrix(c2w, proj_mtx):
    """proj @ w2c with w2c = [R^T | -R^T t] (threestudio/utils/ops.py:281-292).  Pinned by tests/golden/mvp.npz."""
    w2c = torch.zeros(c2w.shape[0], 4, 4).to(c2w)
    w2c[:, :3, :3] = c2w[:, :3, :3].permute(0, 2, 1)
    w2c[:, :3, 3:] = -c2w[:, :3, :3].permute(0, 2, 1) @ c2w[:, :3, 3:]
    w2c[:, 3, 3] = 1.0
    return proj_mtx.to(c2w) @ w2c
