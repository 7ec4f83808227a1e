"""Independent dense float64 PyTorch formulation of the splatting image-formation model.

Purpose: cross-check oracle/raster_oracle.c (forward values and, through autograd, every gradient) with code
that shares nothing with it: no tiles lists, no sorting by key, no hand-derived backward.  Every Gaussian is
evaluated at every pixel; tile culling enters only as a boolean mask.  Small scenes only (memory P x H x W).

Model (constants as in SURVEY.md §2.1): near cull z <= 0.2; EWA cov2D with 1.3 x tanfov clamp and +0.3 px^2
low-pass; radius = ceil(3 sqrt(lambda_max)) with lambda = mid +- sqrt(max(0.1, mid^2 - det)); 16x16 tile
rectangle; power > 0 skipped; alpha = min(0.99, o exp(power)); alpha < 1/255 skipped; stop when
T (1 - alpha) < 1e-4; colour += c alpha T, depth += z alpha T, alpha_out += alpha T; colour += T_final bg.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _eval_sh(deg, sh, dirs):
    # sh [P,M,3], dirs [P,3]
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def _quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    return R


def dense_render(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, H, W, tanfovx, tanfovy,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
                 scale_modifier=1.0, means2D=None):
    """All tensors float64.  viewmatrix / projmatrix are the [4,4] tensors as cameras.py stores them
    (row-vector convention: p_view = [p, 1] @ viewmatrix).  Returns dict(color, depth, alpha, radii, n_contrib)."""
    dt = torch.float64
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], dim=1)
    pv = ph @ viewmatrix
    tz = pv[:, 2]
    valid = tz > 0.2
    phom = ph @ projmatrix
    pw = 1.0 / (phom[:, 3] + 0.0000001)
    ndc = phom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)

    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], dim=-1).reshape(-1, 3, 3)
    else:
        R = _quat_to_rot(rotations)
        L = R * (scale_modifier * scales)[:, None, :]
        Sigma = L @ L.transpose(1, 2)

    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(valid, tz, torch.ones_like(tz))
    tx = torch.clamp(pv[:, 0] / tzs, -limx, limx) * tzs
    ty = torch.clamp(pv[:, 1] / tzs, -limy, limy) * tzs
    zero = torch.zeros_like(tzs)
    J = torch.stack([fx / tzs, zero, -(fx * tx) / (tzs * tzs), zero, fy / tzs, -(fy * ty) / (tzs * tzs)], dim=-1).reshape(-1, 2, 3)
    Rv = viewmatrix[:3, :3].transpose(0, 1)
    Mx = J @ Rv
    cov2 = Mx @ Sigma @ Mx.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    cc = cov2[:, 1, 1] + 0.3
    det = a * cc - b * b
    valid = valid & (det != 0)
    dets = torch.where(valid, det, torch.ones_like(det))
    conA, conB, conC = cc / dets, -b / dets, a / dets
    mid = 0.5 * (a + cc)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
    pd = pix.detach()

    def _toint(v):
        return torch.trunc(v).to(torch.int64)

    rminx = torch.clamp(_toint((pd[:, 0] - radius) / 16), 0, tiles_x)
    rminy = torch.clamp(_toint((pd[:, 1] - radius) / 16), 0, tiles_y)
    rmaxx = torch.clamp(_toint((pd[:, 0] + radius + 15) / 16), 0, tiles_x)
    rmaxy = torch.clamp(_toint((pd[:, 1] + radius + 15) / 16), 0, tiles_y)
    valid = valid & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        col = colors_precomp
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp_min(_eval_sh(sh_degree, shs, d) + 0.5, 0.0)

    # depth order, ties by index (what a stable sort of the (tile|depth) keys gives per tile)
    keyz = torch.where(valid, tz.detach(), torch.full_like(tz, float("inf")))
    order = torch.sort(keyz, stable=True).indices
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxf = xs.reshape(-1).to(dt)
    pyf = ys.reshape(-1).to(dt)
    ptx = (xs.reshape(-1) // 16)
    pty = (ys.reshape(-1) // 16)

    o = order
    dx = pix[o, 0:1] - pxf[None, :]
    dy = pix[o, 1:2] - pyf[None, :]
    power = -0.5 * (conA[o, None] * dx * dx + conC[o, None] * dy * dy) - conB[o, None] * dx * dy
    in_rect = ((ptx[None, :] >= rminx[o, None]) & (ptx[None, :] < rmaxx[o, None]) &
               (pty[None, :] >= rminy[o, None]) & (pty[None, :] < rmaxy[o, None]) & valid[o, None])
    alpha = torch.clamp_max(opacities.reshape(-1)[o, None] * torch.exp(torch.clamp_max(power, 0.0)), 0.99)
    live = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(live, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - alpha
    T_after = torch.cumprod(one_minus, dim=0)
    T_before = torch.cat([torch.ones(1, H * W, dtype=dt), T_after[:-1]], dim=0)
    # termination: first live entry whose test_T < 1e-4 stops the pixel (that entry is NOT blended)
    stop = (live & (T_after.detach() < 0.0001)).to(torch.int64)
    done = torch.cumsum(stop, dim=0) > 0
    alpha = torch.where(done, torch.zeros_like(alpha), alpha)
    live = live & ~done
    one_minus = 1.0 - alpha
    T_after = torch.cumprod(one_minus, dim=0)
    T_before = torch.cat([torch.ones(1, H * W, dtype=dt), T_after[:-1]], dim=0)
    w = alpha * T_before
    color = (w[:, None, :] * col[o][:, :, None]).sum(0) + T_after[-1][None, :] * bg[:, None]
    depth = (w * tz[o, None]).sum(0)
    alpha_out = w.sum(0)
    # n_contrib: 1-based position, within the pixel's TILE list, of the last blended entry
    pos_in_tile = torch.cumsum(in_rect.to(torch.int64), dim=0)
    n_contrib = torch.where(live, pos_in_tile, torch.zeros_like(pos_in_tile)).max(dim=0).values
    return dict(color=color.reshape(3, H, W), depth=depth.reshape(1, H, W), alpha=alpha_out.reshape(1, H, W),
                radii=radii, n_contrib=n_contrib.reshape(H, W), order=order, in_rect=in_rect,
                tiles_touched=torch.where(valid, (rmaxx - rminx) * (rmaxy - rminy), torch.zeros_like(rminx)))


# ----------------------------------------------------------------------------------------------
# scene helpers shared by the tests (pure numpy/torch; follow the reference's conventions)
# ----------------------------------------------------------------------------------------------
def look_at_camera(elev_deg, azim_deg, dist, fovy_deg, H, W, znear=0.01, zfar=100.0):
    """Camera matrices in the layout cameras.py:17-51 produces from a threestudio c2w
    (camera_data.py:448-454).  Returns float64 tensors (viewmatrix, projmatrix(full), campos, tanfovx, tanfovy)."""
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    pos = torch.tensor([dist * math.cos(el) * math.cos(az), dist * math.cos(el) * math.sin(az), dist * math.sin(el)],
                       dtype=torch.float64)
    center = torch.zeros(3, dtype=torch.float64)
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    lookat = (center - pos) / (center - pos).norm()
    right = torch.linalg.cross(lookat, up)
    right = right / right.norm()
    upv = torch.linalg.cross(right, lookat)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, upv, -lookat, pos
    fovy = math.radians(fovy_deg)
    focal = H / (2 * math.tan(fovy / 2))
    fovx = 2 * math.atan(W / (2 * focal))
    w2c = torch.inverse(c2w)
    w2c[1:3, :3] *= -1
    w2c[:3, 3] *= -1
    view = w2c.transpose(0, 1).contiguous()
    tanx, tany = math.tan(fovx / 2), math.tan(fovy / 2)
    Pm = torch.zeros(4, 4, dtype=torch.float64)
    Pm[0, 0] = 1 / tanx
    Pm[1, 1] = 1 / tany
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    proj = Pm.transpose(0, 1)
    full = view @ proj
    campos = torch.inverse(view)[3, :3]
    return view, full, campos, tanx, tany


def random_scene(P, seed, sh_M=1, radius=0.5, scale_lo=0.01, scale_hi=0.08, opa_lo=0.05, opa_hi=0.95):
    g = torch.Generator().manual_seed(seed)
    dt = torch.float64
    d = torch.randn(P, 3, generator=g, dtype=dt)
    d = d / d.norm(dim=1, keepdim=True)
    r = radius * torch.rand(P, 1, generator=g, dtype=dt) ** (1 / 3)
    means = d * r
    scales = scale_lo + (scale_hi - scale_lo) * torch.rand(P, 3, generator=g, dtype=dt)
    q = torch.randn(P, 4, generator=g, dtype=dt)
    q = q / q.norm(dim=1, keepdim=True)
    opa = opa_lo + (opa_hi - opa_lo) * torch.rand(P, 1, generator=g, dtype=dt)
    shs = torch.randn(P, sh_M, 3, generator=g, dtype=dt) * 0.5
    shs[:, 0] += 0.8
    return dict(means3D=means, scales=scales, rotations=q, opacities=opa, shs=shs)
