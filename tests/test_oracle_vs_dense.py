"""The CPU oracle (oracle/raster_oracle.c) against an independent dense float64 autograd model
(tests/dense_reference.py): forward images, integer buffers, and every gradient."""
import numpy as np
import pytest
import torch

from dense_reference import dense_render, look_at_camera, random_scene


def _run(oracle, P, H, W, seed, sh_degree=0, use_cov=False, use_colors=False, bgval=(0.2, 0.5, 0.9), cam=(10.0, 40.0, 1.6, 60.0)):
    M = (sh_degree + 1) ** 2
    sc = random_scene(P, seed, sh_M=M)
    view, proj, campos, tanx, tany = look_at_camera(cam[0], cam[1], cam[2], cam[3], H, W)
    bg = torch.tensor(bgval, dtype=torch.float64)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sc.items()}
    means2D = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    kw = dict(means3D=leaves["means3D"], opacities=leaves["opacities"], viewmatrix=view, projmatrix=proj, campos=campos,
              bg=bg, H=H, W=W, tanfovx=tanx, tanfovy=tany, sh_degree=sh_degree, means2D=means2D)
    okw = dict(image_height=H, image_width=W, tanfovx=tanx, tanfovy=tany, bg=bg.numpy(), scale_modifier=1.0,
               viewmatrix=view.numpy(), projmatrix=proj.numpy(), sh_degree=sh_degree, campos=campos.numpy(),
               means3D=sc["means3D"].numpy(), opacities=sc["opacities"].numpy())
    cov_leaf = col_leaf = None
    if use_cov:
        from dense_reference import _quat_to_rot
        R = _quat_to_rot(sc["rotations"])
        L = R * sc["scales"][:, None, :]
        S = L @ L.transpose(1, 2)
        cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)
        cov_leaf = cov.clone().requires_grad_(True)
        kw["cov3D_precomp"] = cov_leaf
        okw["cov3D_precomp"] = cov.numpy()
    else:
        kw.update(scales=leaves["scales"], rotations=leaves["rotations"])
        okw.update(scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy())
    if use_colors:
        g = torch.Generator().manual_seed(seed + 1)
        col = torch.rand(P, 3, generator=g, dtype=torch.float64)
        col_leaf = col.clone().requires_grad_(True)
        kw["colors_precomp"] = col_leaf
        okw["colors_precomp"] = col.numpy()
    else:
        kw["shs"] = leaves["shs"]
        okw["shs"] = sc["shs"].numpy()
    out = dense_render(**kw)
    ro = oracle.RasterOracle()
    color, radii, depth, alpha = ro.forward(**okw)
    return sc, leaves, means2D, cov_leaf, col_leaf, out, ro, (color, radii, depth, alpha)


@pytest.mark.parametrize("P,H,W,seed,deg", [(64, 32, 48, 1, 0), (150, 64, 64, 2, 1), (100, 40, 56, 3, 3), (200, 64, 80, 4, 2)])
def test_forward_matches_dense(oracle, P, H, W, seed, deg):
    sc, leaves, m2d, _, _, out, ro, (color, radii, depth, alpha) = _run(oracle, P, H, W, seed, sh_degree=deg)
    assert np.array_equal(radii, out["radii"].numpy())
    keys, vals, ranges, tt, nc = ro.binning()
    assert np.array_equal(tt.astype(np.int64), out["tiles_touched"].numpy())
    assert ro.num_rendered == int(out["tiles_touched"].sum())
    np.testing.assert_allclose(color, out["color"].detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(depth, out["depth"].detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(alpha, out["alpha"].detach().numpy(), atol=2e-5)
    assert np.array_equal(nc.astype(np.int64), out["n_contrib"].numpy())
    # per-tile lists: sorted by depth then index; ranges tile the key array
    tiles_x = (W + 15) // 16
    depths = ro.geom()["depths"]
    for t in range(ranges.shape[0]):
        a, b = ranges[t]
        if b > a:
            assert np.all((keys[a:b] >> np.uint64(32)) == t)
            d = depths[vals[a:b]]
            assert np.all(np.diff(d) >= 0)
            same = np.diff(d) == 0
            assert np.all(np.diff(vals[a:b].astype(np.int64))[same] > 0)
    assert int((ranges[:, 1] - ranges[:, 0]).sum()) == ro.num_rendered


def _grad_check(oracle, **kwargs):
    sc, leaves, m2d, cov_leaf, col_leaf, out, ro, (color, radii, depth, alpha) = _run(oracle, **kwargs)
    H, W = color.shape[1:]
    g = torch.Generator().manual_seed(99)
    gC = torch.randn(3, H, W, generator=g, dtype=torch.float64)
    gD = torch.randn(1, H, W, generator=g, dtype=torch.float64)
    gA = torch.randn(1, H, W, generator=g, dtype=torch.float64)
    loss = (out["color"] * gC).sum() + (out["depth"] * gD).sum() + (out["alpha"] * gA).sum()
    loss.backward()
    go = ro.backward(gC.numpy(), gD.numpy(), gA.numpy())

    def close(name, ours, ref, rtol=2e-3):
        ref = ref.detach().numpy()
        scale = np.abs(ref).max() + 1e-12
        err = np.abs(ours - ref).max() / scale
        assert err < rtol, "%s: rel-to-max error %.3e" % (name, err)

    close("means3D", go["means3D"], leaves["means3D"].grad)
    close("means2D", go["means2D"][:, :2], m2d.grad[:, :2])
    close("opacities", go["opacities"], leaves["opacities"].grad)
    if cov_leaf is not None:
        close("cov3D", go["cov3D_precomp"], cov_leaf.grad)
    else:
        close("scales", go["scales"], leaves["scales"].grad)
        close("rotations", go["rotations"], leaves["rotations"].grad)
    if col_leaf is not None:
        close("colors", go["colors_precomp"], col_leaf.grad)
    else:
        close("shs", go["shs"], leaves["shs"].grad)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_backward_matches_autograd_sh(oracle, deg):
    _grad_check(oracle, P=80, H=48, W=48, seed=10 + deg, sh_degree=deg)


def test_backward_matches_autograd_precomp(oracle):
    _grad_check(oracle, P=90, H=48, W=64, seed=21, use_cov=True, use_colors=True)


def test_backward_black_bg_dense_scene(oracle):
    _grad_check(oracle, P=300, H=64, W=64, seed=33, sh_degree=0, bgval=(0.0, 0.0, 0.0), cam=(5.0, 90.0, 1.8, 70.0))


def test_argument_validation(oracle):
    ro = oracle.RasterOracle()
    sc = random_scene(8, 0)
    view, proj, campos, tanx, tany = look_at_camera(0, 0, 2.0, 60, 16, 16)
    with pytest.raises(ValueError):
        ro.forward(image_height=16, image_width=16, tanfovx=tanx, tanfovy=tany, bg=np.zeros(3), scale_modifier=1.0,
                   viewmatrix=view.numpy(), projmatrix=proj.numpy(), sh_degree=0, campos=campos.numpy(),
                   means3D=sc["means3D"].numpy(), opacities=sc["opacities"].numpy(), scales=sc["scales"].numpy(),
                   rotations=sc["rotations"].numpy())  # neither shs nor colors_precomp
