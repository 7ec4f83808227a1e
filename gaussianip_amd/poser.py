"""OpenPose-18 skeleton and its ControlNet pose maps, all views of a step on the GPU.

Mirrors the part of `Skeleton` the training loop uses (threestudio/utils/poser.py): the 18 key points of the default
pose (:665-684, y/z swapped as :694), `scale` (:819-822), the projection + self-occlusion visibility rules
(:836-876) and `openpose_draw` (:832-904).  The reference draws every view with OpenCV on the CPU after a D2H copy of
its mvp matrix and uploads the map again (GaussianIP.py:175-196); here projection and visibility are tensor ops over
the view axis and the canvas is two HIP launches (include/gip_pose.h: OpenCV's ellipse2Poly / fillConvexPoly restated per limb,
then the per-pixel draw), so a step issues no host synchronisation for
its pose maps.  CPU tensors take a numpy path through the same spec (oracle-free: it is the host mirror used by tests).
"""
import ctypes
import math

import numpy as np
import torch

# default pose [18, 3] in [-1, 1]^3 (poser.py:665-684); names / line list poser.py:686-688
_POINTS = [
    [-0.00313026, 0.16587697, 0.05414092], [-0.00857283, 0.1093518, -0.00522604], [-0.06817748, 0.10397182, -0.00657925],
    [-0.11421658, 0.04033477, 0.00040599], [-0.15643744, -0.02915882, 0.03309248], [0.05288884, 0.10729481, -0.00067854],
    [0.10355149, 0.04464601, -0.00735265], [0.15390812, -0.02282556, 0.03085238], [0.03897187, -0.0403506, 0.00220192],
    [0.04027461, -0.15746351, -0.00187036], [0.04605377, -0.26837209, -0.0018945], [-0.0507806, -0.04887162, 0.0022531],
    [-0.04873568, -0.16551849, -0.00128197], [-0.04840493, -0.27510208, -0.00128831], [-0.03098677, 0.19395538, 0.01987491],
    [0.01657042, 0.19560097, 0.02724142], [-0.05411603, 0.17336673, -0.01328044], [0.03733583, 0.16922003, -0.00946565]]
NAMES = ["nose", "neck", "right_shoulder", "right_elbow", "right_wrist", "left_shoulder", "left_elbow", "left_wrist",
         "right_hip", "right_knee", "right_ankle", "left_hip", "left_knee", "left_ankle", "right_eye", "left_eye",
         "right_ear", "left_ear"]
LINES = [[0, 1], [1, 2], [2, 3], [3, 4], [1, 5], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11], [11, 12], [12, 13],
         [0, 14], [14, 16], [0, 15], [15, 17]]


class Skeleton:
    def __init__(self, device="cuda", points3D=None):
        pts = np.asarray(_POINTS if points3D is None else points3D, dtype=np.float32).copy()
        if points3D is None:
            pts[:, [1, 2]] = pts[:, [2, 1]]                     # opengl -> blender (poser.py:694)
        self.device = torch.device(device)
        self._points3D_host = torch.cat([torch.from_numpy(pts), torch.ones(pts.shape[0], 1)], dim=1)              # homogeneous
        self.points3D = self._points3D_host.to(self.device)
        self._lines_host = torch.tensor(LINES, dtype=torch.long)
        self.lines = self._lines_host.to(self.device)
        zoom = torch.zeros(18, dtype=torch.bool)
        zoom[[0, 1, 3, 6, 14, 15, 16, 17]] = True              # key points kept in a head zoom (poser.py:846-849)
        self._zoom_host = zoom
        self._zoom = zoom.to(self.device)
        self.name = list(NAMES)

    def scale(self, delta):
        self.points3D[:, :3] *= 1.1 ** (-delta)
        if self.points3D.data_ptr() != self._points3D_host.data_ptr():
            self._points3D_host[:, :3] *= 1.1 ** (-delta)

    def _on(self, dev):
        """(points3D, lines, zoom) on `dev`: host-side batches (the data module's tensors are CPU tensors) are projected and
        classified on the host — a few hundred flops per view — instead of through ~100 two-microsecond GPU launches."""
        if dev.type == "cpu":
            return self._points3D_host, self._lines_host, self._zoom_host
        return self.points3D, self.lines, self._zoom

    @property
    def hand_centers(self):
        return self.points3D[[self.name.index("left_wrist"), self.name.index("right_wrist")], :3]

    # ------------------------------------------------------------------ projection + visibility (poser.py:836-876)
    def project(self, mvp, H, W):
        """mvp [V,4,4] -> NDC points [V,18,3], pixel xs, ys [V,18]."""
        pts = self._on(mvp.device)[0].to(mvp.dtype) @ mvp.transpose(1, 2)              # [V,18,4]
        ndc = pts[..., :3] / pts[..., 3:]
        return ndc, (ndc[..., 0] + 1) / 2 * W, (ndc[..., 1] + 1) / 2 * H

    def visibility(self, ndc, xs, ys, H, W, azimuth, head_zoom, enable_occlusion=True):
        mask = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
        if not enable_occlusion:
            return mask
        V = mask.shape[0]
        az = torch.as_tensor(azimuth, device=mask.device, dtype=torch.float32).reshape(V)
        hz = torch.as_tensor(head_zoom, device=mask.device, dtype=torch.bool).reshape(V, 1)
        mask = torch.where(hz, self._on(mask.device)[2][None].expand(V, 18), mask).clone()
        mask[:, 16] &= ~((az > 0) & (az < 60))
        mask[:, 17] &= ~((az > 120) & (az < 180))
        z0, z_l, z_r = ndc[:, 0, 2], ndc[:, 17, 2], ndc[:, 16, 2]
        c1 = (z0 > z_l) & (z0 < z_r)                      # right side hidden
        c2 = ~c1 & (z0 < z_l) & (z0 > z_r)                # left side hidden
        c3 = ~c1 & ~c2 & (z0 > z_l) & (z0 > z_r)          # back view
        mask[:, 16] &= ~c1
        mask[:, 14] &= ~(c1 | (c2 & (az < 0) & (az != -180)) | c3)
        mask[:, 15] &= ~((c1 & (az < 0)) | c2 | c3)
        mask[:, 17] &= ~c2
        mask[:, 0] &= ~c3
        return mask

    def limb_parameters(self, xs, ys, mask):
        """[V,17,6] float32: (int centre x, int centre y, int(len/2), drawn?, int angle in degrees, 0) — the arguments of
        cv2.ellipse2Poly at poser.py:889-895."""
        lines = self._on(xs.device)[1]
        X, Y = xs[:, lines], ys[:, lines]                            # [V,17,2]
        mX, mY = X.mean(dim=-1), Y.mean(dim=-1)
        length = ((Y[..., 0] - Y[..., 1]) ** 2 + (X[..., 0] - X[..., 1]) ** 2) ** 0.5
        ang = torch.rad2deg(torch.atan2((Y[..., 0] - Y[..., 1]).double(), (X[..., 0] - X[..., 1]).double())).trunc()
        on = mask[:, lines[:, 0]] & mask[:, lines[:, 1]]
        return torch.stack([mX.trunc().float(), mY.trunc().float(), (length / 2).trunc().float(), on.float(),
                            ang.float(), torch.zeros_like(mX).float()], dim=-1).contiguous()

    # ------------------------------------------------------------------ the drawing call
    def openpose_draw(self, mvp, H, W, azimuth, head_zoom, enable_occlusion=True):
        """mvp [V,4,4] (or [4,4]) -> (canvas [V,H,W,3] float32 in [0,1], all_vis [V] (1 if every key point is drawn),
        xy [V,18,2]).  Same outputs as poser.py:832-904 per view, batched."""
        single = mvp.dim() == 2
        # host-side inputs (mvp, azimuth, head_zoom all CPU / Python values): the small per-view algebra stays on the host
        # and only the three packed parameter arrays of the drawing kernel are uploaded; all_vis / xy then are CPU tensors
        host = mvp.device.type == "cpu" and not (torch.is_tensor(azimuth) and azimuth.is_cuda) and \
            not (torch.is_tensor(head_zoom) and head_zoom.is_cuda)
        mvp = mvp.reshape(-1, 4, 4).to(torch.float32) if host else mvp.reshape(-1, 4, 4).to(self.device, torch.float32, non_blocking=True)
        V = mvp.shape[0]
        ndc, xs, ys = self.project(mvp, H, W)
        mask = self.visibility(ndc, xs, ys, H, W, azimuth, head_zoom, enable_occlusion)
        limbs = self.limb_parameters(xs, ys, mask)
        pts_px = torch.stack([xs.trunc(), ys.trunc()], dim=-1).to(torch.int32).contiguous()
        vis8 = mask.to(torch.uint8).contiguous()
        if host and self.device.type == "cuda":
            pts_px, vis8, limbs = (t.to(self.device, non_blocking=True) for t in (pts_px, vis8, limbs))
        if self.device.type == "cuda":
            from . import _lib
            canvas = torch.empty((V, H, W, 3), dtype=torch.float32, device=self.device)
            lib = _lib.model_lib()
            ws = torch.empty(lib.gip_openpose_workspace_bytes(V, H), dtype=torch.uint8, device=self.device)     # per-row limb spans
            rc = lib.gip_openpose_draw(
                ctypes.c_void_p(pts_px.data_ptr()), ctypes.c_void_p(vis8.data_ptr()), ctypes.c_void_p(limbs.data_ptr()),
                ctypes.c_void_p(canvas.data_ptr()), V, H, W, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("gip_openpose_draw failed with status %d" % rc)
        else:
            raise ValueError("Skeleton.openpose_draw: the canvas is drawn by the HIP kernel; build the skeleton on a GPU")
        all_vis = mask.all(dim=1).to(torch.int64)
        xy = torch.stack([xs, ys], dim=-1)
        if single:
            return canvas[0], all_vis[0], xy[0]
        return canvas, all_vis, xy
