"""OpenPose control maps (include/gip_pose.h): the HIP drawer against oracle/pose_oracle.py (bit-exact, uint8 canvas)
and the batched visibility rules against a per-view loop written like poser.py:843-876."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loop_visibility(ndc, xs, ys, H, W, azimuth, head_zoom):
    mask = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
    if head_zoom:
        mask = np.full_like(mask, False)
        for i in (0, 1, 3, 6, -1, -2, -3, -4):
            mask[i] = True
    if 0 < azimuth < 60:
        mask[-2] = False
    if 120 < azimuth < 180:
        mask[-1] = False
    if ndc[0, 2] > ndc[-1, 2] and ndc[0, 2] < ndc[-2, 2]:
        mask[-2] = False
        mask[-4] = False
        if azimuth < 0:
            mask[-3] = False
    elif ndc[0, 2] < ndc[-1, 2] and ndc[0, 2] > ndc[-2, 2]:
        mask[-1] = False
        mask[-3] = False
        if azimuth < 0 and azimuth != -180:
            mask[-4] = False
    elif ndc[0, 2] > ndc[-1, 2] and ndc[0, 2] > ndc[-2, 2]:
        mask[0] = False
        mask[-3] = False
        mask[-4] = False
    return mask


def test_pose_maps_match_the_oracle_and_the_loop_rules():
    import scenes
    from gaussianip_amd.poser import Skeleton
    from oracle import pose_oracle
    sk = Skeleton("cuda")
    sk.scale(-10)                                           # GaussianIP.py:128
    H = W = 512
    azs = [-170.0, -95.0, -30.0, 10.0, 45.0, 100.0, 150.0, 179.0]
    cams = [scenes.camera(rng_e, az, 1.5, 55.0, H, W) for rng_e, az in zip((-20, 5, 25, 0, -10, 15, 30, -30), azs)]
    mvp = torch.stack([torch.from_numpy(c["projmatrix"]).T for c in cams]).cuda()       # row-vector layout -> column mvp
    head_zoom = [False, False, True, False, False, True, False, False]
    canvas, all_vis, xy = sk.openpose_draw(mvp, H, W, azs, head_zoom)
    assert canvas.shape == (8, H, W, 3) and float(canvas.max()) <= 1.0 and float(canvas.sum()) > 0
    ndc, xs, ys = sk.project(mvp, H, W)
    mask = sk.visibility(ndc, xs, ys, H, W, azs, head_zoom)
    limbs = sk.limb_parameters(xs, ys, mask).cpu().numpy()
    pts_px = torch.stack([xs.trunc(), ys.trunc()], -1).to(torch.int32).cpu().numpy()
    for v in range(8):
        ref_mask = _loop_visibility(ndc[v].cpu().numpy(), xs[v].cpu().numpy(), ys[v].cpu().numpy(), H, W, azs[v], head_zoom[v])
        assert np.array_equal(ref_mask, mask[v].cpu().numpy()), v
        assert int(all_vis[v]) == int(ref_mask.all())
        want = pose_oracle.draw(pts_px[v], ref_mask, limbs[v], H, W)
        got = canvas[v].cpu().numpy()
        assert np.array_equal(np.rint(got * 255).astype(np.uint8), np.rint(want * 255).astype(np.uint8)), v
        assert np.array_equal(got, want), v
    # single-view form, as the reference calls it
    c0, v0, xy0 = sk.openpose_draw(mvp[0], H, W, azs[0], head_zoom[0])
    assert torch.equal(c0, canvas[0]) and int(v0) == int(all_vis[0])


def test_host_side_batch_gives_the_same_pose_maps():
    """mvp / azimuth / head_zoom as CPU tensors (the data module's batch): projection, visibility rules and limb
    parameters run on the host, only the drawing kernel's packed parameters are uploaded — same canvas, same flags."""
    import scenes
    from gaussianip_amd.poser import Skeleton
    sk = Skeleton("cuda")
    sk.scale(-10)
    H = W = 512
    azs = torch.tensor([-170.0, -95.0, 10.0, 45.0, 100.0, 150.0])
    cams = [scenes.camera(e, float(a), 1.5, 55.0, H, W) for e, a in zip((-20, 5, 25, 0, -10, 15), azs)]
    mvp = torch.stack([torch.from_numpy(c["projmatrix"]).T for c in cams])
    hz = torch.tensor([False, False, True, False, True, False])
    c_dev, v_dev, xy_dev = sk.openpose_draw(mvp.cuda(), H, W, azs.cuda(), hz.cuda())
    torch.cuda.set_sync_debug_mode("error")
    try:
        c_host, v_host, xy_host = sk.openpose_draw(mvp, H, W, azs, hz)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert v_host.device.type == "cpu" and c_host.is_cuda
    assert torch.equal(v_host, v_dev.cpu())
    assert torch.equal(c_host, c_dev)


def test_limb_footprints_match_the_opencv_restatement_on_hard_cases():
    """gip_pose_limb_spans_kernel against oracle/pose_oracle.py (ellipse2Poly + fillConvexPoly restated from OpenCV's
    drawing.cpp), bit-exact canvas: every integer angle class, degenerate half-axes (0, 1), long limbs, limbs cut by
    each image border, and the 'not drawn' flag."""
    import ctypes
    from gaussianip_amd import _lib
    from oracle import pose_oracle
    rng = np.random.default_rng(4)
    H, W, V = 96, 128, 12
    limbs = np.zeros((V, 17, 6), np.float32)
    for v in range(V):
        for l in range(17):
            border = rng.random() < 0.35
            cx = rng.integers(-6, W + 6) if border else rng.integers(20, W - 20)
            cy = rng.integers(-6, H + 6) if border else rng.integers(20, H - 20)
            a = [0, 1, 2, 60][rng.integers(0, 4)] if rng.random() < 0.3 else rng.integers(3, 40)
            limbs[v, l] = (cx, cy, a, float(rng.random() < 0.85), rng.integers(-180, 181), 0)
    limbs[0, :8, 4] = [-180, -90, 0, 90, 180, 45, -135, 1]
    pts = np.stack([rng.integers(-3, W + 3, (V, 18)), rng.integers(-3, H + 3, (V, 18))], -1).astype(np.int32)
    vis = (rng.random((V, 18)) < 0.7)
    dev = torch.device("cuda")
    lib = _lib.model_lib()
    t_pts, t_vis, t_l = torch.from_numpy(pts).to(dev), torch.from_numpy(vis.astype(np.uint8)).to(dev), torch.from_numpy(limbs).to(dev)
    canvas = torch.empty((V, H, W, 3), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.gip_openpose_workspace_bytes(V, H), dtype=torch.uint8, device=dev)
    rc = lib.gip_openpose_draw(ctypes.c_void_p(t_pts.data_ptr()), ctypes.c_void_p(t_vis.data_ptr()), ctypes.c_void_p(t_l.data_ptr()),
                               ctypes.c_void_p(canvas.data_ptr()), V, H, W, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                               ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    got = canvas.cpu().numpy()
    spans = ws.view(torch.int32).reshape(V, 17, H).cpu().numpy()
    for v in range(V):
        for l in range(17):                               # the spans themselves, limb by limb
            m = pose_oracle.limb_mask(H, W, *limbs[v, l, [0, 1, 2, 4]]) if limbs[v, l, 3] else np.zeros((H, W), bool)
            lo, hi = spans[v, l] & 0xffff, spans[v, l] >> 16
            mk = (np.arange(W)[None, :] >= lo[:, None]) & (np.arange(W)[None, :] <= hi[:, None]) & ~((lo == 32767) & (hi == 0))[:, None]
            assert np.array_equal(mk, m), (v, l, limbs[v, l])
        want = pose_oracle.draw(pts[v], vis[v], limbs[v], H, W)
        assert np.array_equal(got[v], want), v
    assert lib.gip_openpose_draw(ctypes.c_void_p(t_pts.data_ptr()), ctypes.c_void_p(t_vis.data_ptr()), ctypes.c_void_p(t_l.data_ptr()),
                                 ctypes.c_void_p(canvas.data_ptr()), V, H, W, ctypes.c_void_p(ws.data_ptr()), 16,
                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 2        # workspace too small
