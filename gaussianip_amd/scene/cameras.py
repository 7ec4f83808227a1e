"""Per-view camera matrices in the layout the rasterizer consumes.

Reference: gaussiansplatting/scene/cameras.py:17-51 (Camera), :54-65 (MiniCam).  Pinned by tests/golden/cameras.npz.
Conventions reproduced: FoVx from the vertical focal length and the image WIDTH (:20); world-to-camera =
inverse(c2w) with rows 1-2 of the rotation block and the whole translation negated (threestudio -> 3DGS axes, :23-27);
matrices stored transposed (row-vector convention), full_proj = V^T-stored @ P^T-stored (:48-50); znear 0.01,
zfar 100 (:42-43).  Tensors live on `data_device` (default: the device of `c2w`; the reference forces "cuda").
"""
import torch

from ..utils.graphics import focal2fov, fov2focal, getProjectionMatrix


class Camera(torch.nn.Module):
    def __init__(self, c2w, FoVy, height, width, trans=torch.tensor([0.0, 0.0, 0.0]), scale=1.0, data_device=None):
        super().__init__()
        fovy = float(FoVy)
        dev = c2w.device if data_device is None else torch.device(data_device)
        self.FoVy = fovy
        self.FoVx = focal2fov(fov2focal(fovy, height), width)
        self.image_height, self.image_width = height, width
        self.data_device = dev
        self.znear, self.zfar = 0.01, 100.0
        self.trans, self.scale = trans.float(), scale

        w2c = torch.linalg.inv(c2w.detach().to(torch.float32)).clone()
        w2c[1:3, :3] *= -1
        w2c[:3, 3] *= -1
        # all 4x4 algebra runs where c2w lives and only the results move to `data_device`: with host-side camera
        # parameters (the data module's) a step then issues no device-side inverse and no host synchronisation
        wvt = w2c.t().contiguous().float()
        proj = getProjectionMatrix(self.znear, self.zfar, self.FoVx, self.FoVy).t().float().to(wvt.device)
        full = (wvt @ proj).float()
        center = torch.linalg.inv(wvt)[3, :3].float()
        self.world_view_transform = wvt.to(dev, non_blocking=True)
        self.projection_matrix = proj.to(dev, non_blocking=True)
        self.full_proj_transform = full.to(dev, non_blocking=True)
        self.camera_center = center.to(dev, non_blocking=True)


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx = fovy, fovx
        self.znear, self.zfar = znear, zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        self.camera_center = torch.linalg.inv(world_view_transform)[3][:3]
