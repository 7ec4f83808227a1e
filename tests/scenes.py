"""Seeded synthetic scenes shared by the parity tests and bench.py.

cfg1: P random Gaussians in a ball of radius 0.5 (recipe of gs_renderer.py:900-908), isotropic scale from the mean
3-NN distance (gaussian_model.py:123-124), identity rotation, opacity 0.1 (:128), colour 0.5 -> f_dc = 0
(GaussianIP.py:139, sh_utils.py:114); camera elevation 5, azimuth 90, distance 1.8, fovy 70 (configs/exp.yaml:37-40).
"stress": opacity U(0.02, 0.98), anisotropic scales x U(0.3, 3), random unit quaternions, colours U(0, 1).
"human": points on capsules around the 17 OpenPose limbs built from the 18 keypoints of poser.py:665-684 (a synthetic
stand-in for the licensed SMPL-X surface), height 1.556.
"""
import math

import numpy as np

# 18 OpenPose keypoints (x right, y up, z forward), unit-height A-pose stand-in; limbs as in the OpenPose skeleton
_KP = np.array([
    [0.00, 0.86, 0.02], [0.00, 0.76, 0.00], [-0.11, 0.76, 0.00], [-0.21, 0.62, 0.00], [-0.30, 0.48, 0.02],
    [0.11, 0.76, 0.00], [0.21, 0.62, 0.00], [0.30, 0.48, 0.02], [-0.06, 0.47, 0.00], [-0.08, 0.25, 0.01],
    [-0.09, 0.03, 0.00], [0.06, 0.47, 0.00], [0.08, 0.25, 0.01], [0.09, 0.03, 0.00], [-0.02, 0.89, 0.04],
    [0.02, 0.89, 0.04], [-0.05, 0.88, 0.00], [0.05, 0.88, 0.00]], dtype=np.float64)
_LIMBS = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12), (12, 13),
          (1, 0), (0, 14), (14, 16), (0, 15), (15, 17)]
_RADII = [0.035, 0.035, 0.028, 0.024, 0.028, 0.024, 0.075, 0.045, 0.035, 0.075, 0.045, 0.035, 0.03, 0.05, 0.045,
          0.05, 0.045]


def knn_scale(points, k=3, chunk=2048):
    """mean squared distance to the 3 nearest neighbours, float32 (simple_knn.cu:147-183 semantics), via numpy."""
    pts = points.astype(np.float32)
    P = pts.shape[0]
    out = np.zeros(P, np.float32)
    try:
        from scipy.spatial import cKDTree
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=k + 1)
        out = (d[:, 1:] ** 2).mean(axis=1).astype(np.float32)
    except Exception:
        for s in range(0, P, chunk):
            d2 = ((pts[s:s + chunk, None, :] - pts[None, :, :]) ** 2).sum(-1)
            d2[np.arange(min(chunk, P - s)), np.arange(s, min(P, s + chunk))] = np.inf
            out[s:s + chunk] = np.sort(d2, axis=1)[:, :k].mean(axis=1)
    return out


def ball_points(P, rng, radius=0.5):
    phis = rng.random(P) * 2 * np.pi
    costheta = rng.random(P) * 2 - 1
    thetas = np.arccos(costheta)
    mu = rng.random(P)
    r = radius * np.cbrt(mu)
    return np.stack([r * np.sin(thetas) * np.cos(phis), r * np.sin(thetas) * np.sin(phis), r * np.cos(thetas)], 1)


def human_points(P, rng, height=1.556):
    seg_len = np.array([np.linalg.norm(_KP[a] - _KP[b]) for a, b in _LIMBS])
    area = seg_len * np.array(_RADII) + 2 * np.array(_RADII) ** 2
    counts = rng.multinomial(P, area / area.sum())
    pts = []
    for (a, b), r, n in zip(_LIMBS, _RADII, counts):
        t = rng.random(n)
        axis = _KP[b] - _KP[a]
        base = _KP[a][None, :] + t[:, None] * axis[None, :]
        d = rng.normal(size=(n, 3))
        ax = axis / (np.linalg.norm(axis) + 1e-12)
        d -= (d @ ax)[:, None] * ax[None, :]
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-12
        pts.append(base + r * d)
    p = np.concatenate(pts, 0)
    p = p[:, [0, 2, 1]]  # y/z swap: z up (poser.py:694)
    p[:, 2] -= 0.45
    return p * height


def make_scene(kind, P, seed=42, sh_degree=0):
    """Returns float32 numpy arrays: means3D [P,3], scales [P,3], rotations [P,4], opacities [P,1], shs [P,M,3]."""
    rng = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2
    if kind in ("ball", "stress"):
        xyz = ball_points(P, rng)
    elif kind == "human":
        xyz = human_points(P, rng)
    else:
        raise ValueError(kind)
    d2 = np.maximum(knn_scale(xyz), 1e-7)
    scales = np.repeat(np.sqrt(d2)[:, None], 3, 1).astype(np.float32)
    rots = np.zeros((P, 4), np.float32)
    rots[:, 0] = 1
    opac = np.full((P, 1), 0.1, np.float32)
    shs = np.zeros((P, M, 3), np.float32)
    if kind == "stress":
        opac = rng.uniform(0.02, 0.98, (P, 1)).astype(np.float32)
        scales = (scales * rng.uniform(0.3, 3.0, (P, 3))).astype(np.float32)
        q = rng.normal(size=(P, 4))
        rots = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        shs[:, 0, :] = ((rng.uniform(0, 1, (P, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
        if M > 1:
            shs[:, 1:, :] = rng.normal(size=(P, M - 1, 3)).astype(np.float32) * 0.1
    return dict(means3D=xyz.astype(np.float32), scales=scales, rotations=rots, opacities=opac, shs=shs)


def camera(elev_deg, azim_deg, dist, fovy_deg, H, W):
    """float32 numpy camera in the layout cameras.py:17-51 produces.  Returns dict(viewmatrix [4,4], projmatrix [4,4],
    campos [3], tanfovx, tanfovy)."""
    from dense_reference import look_at_camera
    view, full, campos, tanx, tany = look_at_camera(elev_deg, azim_deg, dist, fovy_deg, H, W)
    return dict(viewmatrix=view.numpy().astype(np.float32), projmatrix=full.numpy().astype(np.float32),
                campos=campos.numpy().astype(np.float32), tanfovx=float(tanx), tanfovy=float(tany))


def trained_look(sc, seed=7):
    """A densified / optimised look of a scene made by make_scene (in place, returned): opaque (0.6), larger anisotropic
    splats (1-3x per axis) with arbitrary orientation and colour.  Exercises early termination (T < 1e-4), the 0.99 alpha
    cap and long occupied tile lists — the state a training run spends most of its time in, next to the init state §8d
    prescribes for the headline number."""
    rng = np.random.default_rng(seed)
    P = sc["means3D"].shape[0]
    sc["scales"] = (sc["scales"] * rng.uniform(1.0, 3.0, (P, 3))).astype(np.float32)
    sc["opacities"][:] = 0.6
    q = rng.normal(size=(P, 4))
    sc["rotations"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    sc["shs"][:, 0, :] = ((rng.uniform(0, 1, (P, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
    return sc


def train_cameras(n, seed, H, W):
    """n cameras from the training ranges (configs/exp.yaml:6-45; camera_data.py:336-363): elevation +-30,
    batch-uniform azimuth, distance 1.3-1.7, fovy 40-70."""
    rng = np.random.default_rng(seed)
    az0 = rng.uniform(-180, 180)
    cams = []
    for i in range(n):
        el = rng.uniform(-30, 30)
        az = az0 + 360.0 * i / n
        cams.append(camera(el, az, rng.uniform(1.3, 1.7), rng.uniform(40, 70), H, W))
    return cams


def orbit_c2w(elev_deg, azim_deg, dist, center_z=0.0):
    """threestudio camera-to-world [4,4] torch float32 (camera_data.py:423-454): z up, camera on the sphere of radius
    `dist` around (0, 0, center_z), looking at that centre."""
    import torch
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    center = torch.tensor([0.0, 0.0, float(center_z)])
    pos = torch.tensor([dist * math.cos(el) * math.cos(az), dist * math.cos(el) * math.sin(az), dist * math.sin(el)]) + center
    up = torch.tensor([0.0, 0.0, 1.0])
    lookat = torch.nn.functional.normalize(center - pos, dim=-1)
    right = torch.nn.functional.normalize(torch.linalg.cross(lookat, up), dim=-1)
    upv = torch.nn.functional.normalize(torch.linalg.cross(right, lookat), dim=-1)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, upv, -lookat, pos
    return c2w


def train_batch(rng, B=4, H=1024, W=1024, device=None):
    """One batch of the random-camera data module (camera_data.py:330-470 with the ranges of configs/exp.yaml:6-45):
    elevation U(-30, 30), batch-uniform azimuth over (-180, 180), distance U(1.3, 1.7), fovy U(40, 70) degrees, looking
    at the origin.  Camera algebra stays on the host (c2w, fovy, mvp_mtx: CPU tensors, like the data module's), the
    per-view scalars the prompt lookup reads go to `device`.  Keys as the reference's batch dict."""
    import torch
    from gaussianip_amd.utils.graphics import get_mvp_matrix, get_projection_matrix
    el = rng.uniform(-30, 30, B).astype(np.float32)
    az = (((rng.random(B) + np.arange(B)) / B) * 360.0 - 180.0).astype(np.float32)
    dist = rng.uniform(1.3, 1.7, B).astype(np.float32)
    fovy = np.radians(rng.uniform(40, 70, B)).astype(np.float32)
    c2w = torch.stack([orbit_c2w(float(el[k]), float(az[k]), float(dist[k])) for k in range(B)])
    fovy_t = torch.from_numpy(fovy)
    mvp = get_mvp_matrix(c2w, get_projection_matrix(fovy_t, W / H, 0.1, 1000.0))
    dev = (lambda t: t.to(device, non_blocking=True)) if device is not None else (lambda t: t)
    return dict(c2w=c2w, fovy=fovy_t, mvp_mtx=mvp, height=H, width=W, elevation=dev(torch.from_numpy(el)),
                azimuth=dev(torch.from_numpy(az)), center=dev(torch.zeros(B)), camera_distances=dev(torch.from_numpy(dist)))
