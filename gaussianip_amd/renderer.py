"""render() and friends: the reference's renderer entry points on top of the HIP rasterizer.

Reference: gaussiansplatting/gaussian_renderer/__init__.py — render :18-104, render_with_smaller_scale :106-193
(identical body; used for val/test at threestudio/systems/GaussianIP.py:156-157), render_deformed :195-265.
Same arguments, same returned dict keys / shapes / dtypes:
    render, viewspace_points, visibility_filter (radii > 0), radii, depth_3dgs, alpha_3dgs.

Deviations, all documented in DESIGN.md:
  * the reference casts the `None` placeholders with `.float()` (:88,:91,:92), so its convert_SHs_python /
    compute_cov3D_python switches can only work through gs_renderer.Renderer.render (gs_renderer.py:969-1001);
    here both switches work (None inputs are passed through un-cast, as gs_renderer does);
  * `render_views` renders all cameras of a batch in ONE launch set (the reference loops, GaussianIP.py:154-173).
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
from .utils.sh import eval_sh


def _settings(cam, pc, bg_color, scaling_modifier):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)


def _select_inputs(cam, pc, pipe, scaling_modifier, override_color, activated=False):
    """(scales, rotations, cov3D_precomp, shs, colors_precomp) following the reference's switches (:57-80).
    `activated`: the caller also needs the opacity and takes all three activations from one launch (GaussianModel.get_activated),
    left in pc._act_cache."""
    scales = rotations = cov3D = shs = colors = None
    if pipe.compute_cov3D_python:
        cov3D = pc.get_covariance(scaling_modifier)
    elif activated and hasattr(pc, "get_activated"):
        pc._act_cache = pc.get_activated()
        _, scales, rotations = pc._act_cache
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    if override_color is not None:
        colors = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        d = pc.get_xyz - cam.camera_center.repeat(feats.shape[0], 1)
        d = d / d.norm(dim=1, keepdim=True)
        colors = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, d) + 0.5, 0.0)
    else:
        shs = pc.get_features
    return scales, rotations, cov3D, shs, colors


def _f(t):
    return None if t is None else t.float()


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render one camera.  `bg_color` must live on the GPU."""
    xyz = pc.get_xyz
    # zero tensor whose .grad receives the screen-space (NDC) gradient of the 2-D means: the densification signal
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    scales, rotations, cov3D, shs, colors = _select_inputs(viewpoint_camera, pc, pipe, scaling_modifier, override_color)
    image, radii, depth, alpha = rasterizer(
        means3D=xyz.float(), means2D=screenspace_points.float(), shs=_f(shs), colors_precomp=colors,
        opacities=pc.get_opacity.float(), scales=_f(scales), rotations=_f(rotations), cov3D_precomp=cov3D)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "depth_3dgs": depth, "alpha_3dgs": alpha}


def render_with_smaller_scale(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
                              override_color=None):
    """Same body as render() in the reference (:106-193); kept as a separate name for the val/test call site."""
    return render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)


def render_deformed(viewpoint_camera, means3D, opacity, scales, rotations, shs, active_sh_degree, bg_color,
                    scaling_modifier=1.0):
    """Explicit-tensor variant (:195-265): returns no depth / alpha entries."""
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True, device=means3D.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    cam = viewpoint_camera
    st = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=active_sh_degree,
        campos=cam.camera_center, prefiltered=False, debug=False)
    image, radii, _depth, _alpha = GaussianRasterizer(raster_settings=st)(
        means3D=means3D.float(), means2D=screenspace_points.float(), shs=shs.float(), colors_precomp=None,
        opacities=opacity.float(), scales=scales.float(), rotations=rotations.float())
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


_ZEROS = {}


class _LazyDict(dict):
    """A dict whose listed entries are computed on first access (the reference's return dicts carry tensors a training step
    never reads: `visibility_filter` per view, the activated `scale`)."""

    def __init__(self, items, lazy):
        super().__init__(items)
        self._lazy = dict(lazy)

    def __missing__(self, key):
        fn = self._lazy.pop(key, None)
        if fn is None:
            raise KeyError(key)
        v = self[key] = fn()
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default


def render_views(cameras, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """All cameras of a batch in one launch set.  Returns the reference dict with a leading view axis:
    render [V,3,H,W], viewspace_points [V,P,3] (grad carrier), visibility_filter / radii [V,P],
    depth_3dgs / alpha_3dgs [V,1,H,W].  Per-view results equal render() called camera by camera."""
    xyz = pc.get_xyz
    V = len(cameras)
    # the gradient carrier of the 2-D means (the reference's `zeros_like(...) + 0` with retain_grad): a fresh autograd leaf per call
    # over ONE cached block of zeros per shape — nothing ever writes the values, only .grad is read — instead of a fill and an add
    # of V x P x 3 floats per step
    key = (V,) + tuple(xyz.shape) + (xyz.dtype, xyz.device)
    zeros = _ZEROS.get(key)
    if zeros is None:
        _ZEROS.clear()
        zeros = _ZEROS[key] = torch.zeros((V,) + tuple(xyz.shape), dtype=xyz.dtype, device=xyz.device)
    screenspace_points = zeros.detach().requires_grad_(True)
    if pipe.convert_SHs_python and override_color is None:
        raise ValueError("render_views: convert_SHs_python colours are view dependent; use render() per camera")
    scales, rotations, cov3D, shs, colors = _select_inputs(cameras[0], pc, pipe, scaling_modifier, override_color, activated=True)
    act = pc.__dict__.pop("_act_cache", None)
    opacity = act[0] if act is not None else pc.get_opacity
    sts = [_settings(c, pc, bg_color, scaling_modifier) for c in cameras]
    image, radii, depth, alpha = rasterize_views(
        xyz.float(), screenspace_points.float(), opacity.float(), sts, shs=_f(shs), colors_precomp=colors,
        scales=_f(scales), rotations=_f(rotations), cov3D_precomp=cov3D)
    return _LazyDict({"render": image, "viewspace_points": screenspace_points, "radii": radii, "depth_3dgs": depth, "alpha_3dgs": alpha},
                     {"visibility_filter": lambda r_=radii: r_ > 0})
