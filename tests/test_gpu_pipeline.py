"""GPU tests of the callers either side of the rasterizer: distCUDA2 replacement, GaussianModel on device, render() /
render_views(), the Python pre-stage switches (SURVEY.md §4 self-consistency items 1-3) and densification."""
import math
import os
from argparse import ArgumentParser

import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_knn_matches_oracle_bit_exact(oracle):
    from gaussianip_amd.knn import distCUDA2
    d = np.load(os.path.join(GOLD, "knn_dist2.npz"))
    pts = d["points"]
    out = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    assert np.array_equal(out, oracle.knn_mean_dist2(pts))
    np.testing.assert_allclose(out, d["dist2"], rtol=2e-5)
    rng = np.random.default_rng(3)
    big = rng.normal(size=(20011, 3)).astype(np.float32)   # ragged size (not a multiple of the tile)
    oracle.set_threads(8)
    ref = oracle.knn_mean_dist2(big)
    oracle.set_threads(1)
    assert np.array_equal(distCUDA2(torch.from_numpy(big).cuda()).cpu().numpy(), ref)
    dup = np.zeros((5, 3), np.float32)                      # coincident points: distances 0
    assert float(distCUDA2(torch.from_numpy(dup).cuda()).abs().max()) == 0.0


def test_box_pruned_knn_equals_all_pairs_bit_for_bit(oracle):
    """simple_knn's scheme (Morton sort, 1024-point boxes that prune the exact search: simple_knn.cu:45-185) against the tiled
    all-pairs kernel and the CPU oracle: the identical float per point — small ragged clouds, coincident points, a degenerate
    (planar) cloud, the shipped 100k human surface, and 1M points (the size of BASELINE configs[4]) against sampled brute force."""
    import time
    from gaussianip_amd.knn import distCUDA2
    import scenes
    rng = np.random.default_rng(5)
    big = rng.normal(size=(20011, 3)).astype(np.float32)
    oracle.set_threads(8)
    ref = oracle.knn_mean_dist2(big)
    oracle.set_threads(1)
    for mode in ("all_pairs", "box_pruned"):
        assert np.array_equal(distCUDA2(torch.from_numpy(big).cuda(), mode).cpu().numpy(), ref), mode
    for pts in (np.zeros((5, 3), np.float32), rng.normal(size=(3, 3)).astype(np.float32), rng.normal(size=(1025, 3)).astype(np.float32),
                np.concatenate([rng.normal(size=(3000, 2)), np.zeros((3000, 1))], axis=1).astype(np.float32),       # planar: one Morton axis degenerate
                np.repeat(rng.normal(size=(700, 3)), 3, axis=0).astype(np.float32)):                                 # every point three times
        t = torch.from_numpy(pts).cuda()
        assert torch.equal(distCUDA2(t, "box_pruned"), distCUDA2(t, "all_pairs")), pts.shape
    human = torch.from_numpy(scenes.human_points(100000, np.random.default_rng(42)).astype(np.float32)).cuda()
    a = distCUDA2(human, "all_pairs")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = distCUDA2(human, "box_pruned")
    torch.cuda.synchronize()
    t_pruned = time.perf_counter() - t0
    assert torch.equal(a, b) and torch.equal(distCUDA2(human), b)             # "auto" takes the pruned path above 32 768 points
    million = torch.from_numpy(scenes.human_points(1000000, np.random.default_rng(43)).astype(np.float32)).cuda()
    t0 = time.perf_counter()
    c = distCUDA2(million)
    torch.cuda.synchronize()
    t_million = time.perf_counter() - t0
    sel = torch.from_numpy(np.random.default_rng(1).choice(1000000, 2048, replace=False)).cuda()
    d2 = torch.cdist(million[sel].double(), million.double()) ** 2
    d2[torch.arange(2048, device="cuda"), sel] = float("inf")
    brute = d2.topk(3, largest=False).values.float().mean(1)
    assert float(((c[sel] - brute).abs() / brute.clamp_min(1e-12)).max()) < 1e-4
    print("box-pruned k-NN: 100k points %.1f ms, 1M points %.1f ms" % (t_pruned * 1e3, t_million * 1e3))


def _model(P=5000, seed=7, sh_degree=0):
    from gaussianip_amd.arguments import OptimizationParams
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.utils import BasicPointCloud
    rng = np.random.default_rng(seed)
    pts = scenes.human_points(P, rng).astype(np.float32)
    cols = rng.uniform(0.2, 0.8, (P, 3)).astype(np.float32)
    gm = GaussianModel(sh_degree)
    gm.create_from_pcd(BasicPointCloud(pts, cols, None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    return gm


def _camera(el, az, dist, fovy_deg, H, W):
    from gaussianip_amd.scene import Camera
    import sys
    from scenes import orbit_c2w
    return Camera(c2w=orbit_c2w(el, az, dist).cuda(), FoVy=math.radians(fovy_deg), height=H, width=W)


def test_render_matches_oracle_and_feeds_densification(oracle):
    from gaussianip_amd.arguments import PipelineParams
    from gaussianip_amd.renderer import render
    gm = _model()
    cam = _camera(10.0, 30.0, 1.6, 55.0, 160, 128)
    bg = torch.tensor([0.0, 0.0, 0.0], device="cuda")
    pkg = render(cam, gm, PipelineParams(ArgumentParser()), bg)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth_3dgs", "alpha_3dgs"}
    assert pkg["render"].shape == (3, 160, 128) and pkg["depth_3dgs"].shape == (1, 160, 128)
    assert pkg["radii"].dtype == torch.int32 and pkg["visibility_filter"].dtype == torch.bool
    ro = oracle.RasterOracle()
    o_color, o_radii, o_depth, o_alpha = ro.forward(
        image_height=160, image_width=128, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=np.zeros(3, np.float32), scale_modifier=1.0, viewmatrix=cam.world_view_transform.cpu().numpy(),
        projmatrix=cam.full_proj_transform.cpu().numpy(), sh_degree=0, campos=cam.camera_center.cpu().numpy(),
        means3D=gm.get_xyz.detach().cpu().numpy(), opacities=gm.get_opacity.detach().cpu().numpy(),
        shs=gm.get_features.detach().cpu().numpy(), scales=gm.get_scaling.detach().cpu().numpy(),
        rotations=gm.get_rotation.detach().cpu().numpy())
    assert np.array_equal(pkg["radii"].cpu().numpy(), o_radii)
    np.testing.assert_allclose(pkg["render"].detach().cpu().numpy(), o_color, atol=1e-4)
    np.testing.assert_allclose(pkg["depth_3dgs"].detach().cpu().numpy(), o_depth, atol=2e-4)
    # loss like threestudio/systems/GaussianIP.py:225,382-384: sparsity on depth / depth.max()
    depth = pkg["depth_3dgs"]
    opacity = depth / (depth.max() + 1e-5)
    loss = pkg["render"].mean() + torch.sqrt(opacity ** 2 + 0.01).mean()
    loss.backward()
    vs = pkg["viewspace_points"]
    assert vs.grad is not None and vs.grad.shape == (gm.get_xyz.shape[0], 3) and float(vs.grad[:, 2].abs().max()) == 0.0
    for p in (gm._xyz, gm._features_dc, gm._scaling, gm._rotation, gm._opacity):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0
    # densification bookkeeping + an optimizer step + densify keep everything renderable
    gm.max_radii2D = torch.max(gm.max_radii2D, pkg["radii"].float())
    gm.add_densification_stats(vs.grad, pkg["visibility_filter"])
    gm.optimizer.step()
    n0 = gm.get_xyz.shape[0]
    gm.densify_and_prune(1e-7, 0.05, 4.0, None, 0.015)
    assert gm.get_xyz.shape[0] != n0
    pkg2 = render(cam, gm, PipelineParams(ArgumentParser()), bg)
    assert pkg2["radii"].shape[0] == gm.get_xyz.shape[0] and torch.isfinite(pkg2["render"]).all()


def test_python_prestage_switches_agree_with_in_kernel_path():
    """convert_SHs_python / compute_cov3D_python must reproduce the in-kernel SH and cov3D."""
    from gaussianip_amd.arguments import PipelineParams
    from gaussianip_amd.renderer import render
    gm = _model(P=3000, seed=11, sh_degree=2)
    gm.active_sh_degree = 2
    with torch.no_grad():
        gm._features_rest.add_(torch.randn_like(gm._features_rest) * 0.05)
        gm._rotation.add_(torch.randn_like(gm._rotation) * 0.3)
        gm._scaling.add_(torch.randn_like(gm._scaling) * 0.3)
    cam = _camera(-5.0, 140.0, 1.5, 60.0, 128, 128)
    bg = torch.tensor([0.2, 0.3, 0.4], device="cuda")
    base = render(cam, gm, PipelineParams(ArgumentParser()), bg)
    for kw in (dict(convert_SHs_python=True), dict(compute_cov3D_python=True),
               dict(convert_SHs_python=True, compute_cov3D_python=True)):
        alt = render(cam, gm, PipelineParams(ArgumentParser(), **kw), bg)
        assert (alt["radii"] - base["radii"]).abs().max() <= 1      # a radius may flip by one on a ceil() boundary
        assert float((alt["render"] - base["render"]).abs().max()) < 2e-3
        assert float((alt["alpha_3dgs"] - base["alpha_3dgs"]).abs().max()) < 2e-3
    ovr = render(cam, gm, PipelineParams(ArgumentParser()), bg, override_color=torch.full((3000, 3), 0.25, device="cuda"))
    a = ovr["alpha_3dgs"]
    assert torch.allclose(ovr["render"], 0.25 * a + (1 - a) * bg[:, None, None], atol=2e-3)


def test_render_views_equals_per_camera_render():
    from gaussianip_amd.arguments import PipelineParams
    from gaussianip_amd.renderer import render, render_views
    gm = _model(P=4000, seed=5)
    cams = [_camera(5.0, 90.0 * i, 1.5, 50.0, 96, 96) for i in range(4)]
    bg = torch.zeros(3, device="cuda")
    pipe = PipelineParams(ArgumentParser())
    batch = render_views(cams, gm, pipe, bg)
    (batch["render"].square().mean() + batch["depth_3dgs"].mean()).backward()
    g_batch = gm._xyz.grad.clone()
    vs_batch = batch["viewspace_points"].grad.clone()
    gm._xyz.grad = None
    singles = [render(c, gm, pipe, bg) for c in cams]
    (torch.stack([s["render"] for s in singles]).square().mean() + torch.stack([s["depth_3dgs"] for s in singles]).mean()).backward()
    for i, s in enumerate(singles):
        assert torch.equal(s["render"], batch["render"][i]) and torch.equal(s["radii"], batch["radii"][i])
        assert torch.equal(s["viewspace_points"].grad, vs_batch[i])
    assert float((gm._xyz.grad - g_batch).abs().max() / gm._xyz.grad.abs().max()) < 1e-5


def test_dropin_package_names_resolve():
    import gaussianip_amd
    gaussianip_amd.install_dropin()
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
    from simple_knn._C import distCUDA2
    assert float(distCUDA2(torch.rand(100, 3).cuda()).min()) > 0


def test_stage_one_step_schedule_and_loss():
    """GaussianIP.forward / training_step / on_before_optimizer_step mirror (gaussianip_amd/system.py)."""
    import sys
    from gaussianip_amd.arguments import PipelineParams
    from gaussianip_amd.system import StageOneStep
    from scenes import orbit_c2w
    gm = _model(P=6000, seed=3)
    st = StageOneStep(gm, PipelineParams(ArgumentParser()), torch.zeros(3, device="cuda"))
    batch = dict(c2w=torch.stack([orbit_c2w(5.0, 90.0 * i, 1.5) for i in range(4)]).cuda(),
                 fovy=torch.full((4,), math.radians(55.0)), height=128, width=128)
    fired = {}
    for step in (499, 500, 1000, 1500, 1700, 1800):
        out = st.forward(batch)
        assert out["comp_rgb"].shape == (4, 128, 128, 3) and out["opacity"].shape == (4, 128, 128, 1)
        assert float(out["opacity"].max()) <= 1.0
        loss = st.loss(out, {"loss_sds": (out["comp_rgb"] ** 2).mean()})
        gm.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        n0 = gm.get_xyz.shape[0]
        fired[step] = st.on_before_optimizer_step(step)
        if fired[step] is None:
            gm.optimizer.step()
        else:
            assert gm.get_xyz.shape[0] != n0 or fired[step] == "prune_only"
    assert fired == {499: None, 500: "densify_and_prune", 1000: "densify_and_prune", 1500: "densify_and_prune",
                     1700: None, 1800: "prune_only"}


def test_fused_densify_matches_stepwise_reference_semantics():
    """densify_and_prune through ONE gip_gather_rows launch (include/gip_model.h) against the step-by-step
    mask / cat path that mirrors gaussian_model.py:357-411: same survivors, same order, same Adam moments — bitwise."""
    from argparse import ArgumentParser
    import copy
    from gaussianip_amd.arguments import OptimizationParams
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.utils import BasicPointCloud

    def build():
        rng = np.random.default_rng(7)
        P = 20000
        pts = rng.normal(0, 0.3, (P, 3)).astype(np.float32)
        gm = GaussianModel(1)
        gm.create_from_pcd(BasicPointCloud(pts, rng.uniform(0, 1, (P, 3)).astype(np.float32), None), 4.0)
        gm.training_setup(OptimizationParams(ArgumentParser()))
        g = torch.Generator(device="cuda").manual_seed(3)
        with torch.no_grad():
            gm._scaling.add_(torch.randn(gm._scaling.shape, device="cuda", generator=g) * 0.8)
            gm._opacity.add_(torch.randn(gm._opacity.shape, device="cuda", generator=g) * 2.0)
            gm._rotation.copy_(torch.randn(gm._rotation.shape, device="cuda", generator=g))
        for grp in gm.optimizer.param_groups:          # one Adam step so that the moments are non-trivial
            p = grp["params"][0]
            p.grad = torch.randn(p.shape, device="cuda", generator=g) * 1e-3
        gm.optimizer.step()
        gm.xyz_gradient_accum = torch.rand((P, 1), device="cuda", generator=g) * 4e-3
        gm.denom = torch.randint(0, 3, (P, 1), device="cuda", generator=g).float()      # zeros -> NaN -> 0 path
        gm.max_radii2D = torch.rand((P,), device="cuda", generator=g) * 100
        return gm

    args = dict(max_grad=1e-3, min_opacity=0.05, extent=4.0, max_screen_size=20, max_world_size=0.5)
    a, b = build(), build()
    torch.manual_seed(11)
    a._densify_and_prune_stepwise(**args)
    torch.manual_seed(11)
    b.densify_and_prune(**args)
    assert a.get_xyz.shape[0] == b.get_xyz.shape[0] and a.get_xyz.shape[0] != 20000
    for attr in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(a, attr), getattr(b, attr)), attr
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        sa, sb = a.optimizer.state[ga["params"][0]], b.optimizer.state[gb["params"][0]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), ga["name"]
        assert ga["params"][0] is getattr(a, dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling", rotation="_rotation")[ga["name"]])
    # the rebuilt model trains on: one more optimizer step works and keeps shapes
    for grp in b.optimizer.param_groups:
        p = grp["params"][0]
        p.grad = torch.zeros_like(p)
    b.optimizer.step()


def test_spatial_sort_is_invisible_to_the_renderer():
    """GaussianModel.sort_spatially(): parameters, Adam moments and statistics are permuted consistently; render() output
    changes only by float summation order and by the order of entries with IDENTICAL depth bits (the reference algorithm
    breaks depth ties by Gaussian index, so a few pixels per image move by one entry's contribution)."""
    from argparse import ArgumentParser
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.renderer import render
    from gaussianip_amd.scene import Camera, GaussianModel
    from gaussianip_amd.utils import BasicPointCloud
    rng = np.random.default_rng(3)
    P = 30000
    pts = scenes.human_points(P, rng).astype(np.float32)[rng.permutation(P)]
    gm = GaussianModel(0)
    gm.create_from_pcd(BasicPointCloud(pts, rng.uniform(0, 1, (P, 3)).astype(np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    for grp in gm.optimizer.param_groups:
        grp["params"][0].grad = torch.randn_like(grp["params"][0]) * 1e-4
    gm.optimizer.step()
    cam = _camera(10.0, 30.0, 1.6, 55.0, 256, 256)
    pipe = PipelineParams(ArgumentParser())
    bg = torch.zeros(3, device="cuda")
    before = render(cam, gm, pipe, bg)["render"].detach().clone()
    xyz0 = gm.get_xyz.detach().clone()
    m0 = gm.optimizer.state[gm.optimizer.param_groups[0]["params"][0]]["exp_avg"].clone()
    perm = gm.sort_spatially()
    assert torch.equal(gm.get_xyz.detach(), xyz0[perm])
    assert torch.equal(gm.optimizer.state[gm.optimizer.param_groups[0]["params"][0]]["exp_avg"], m0[perm])
    assert gm.optimizer.param_groups[0]["params"][0] is gm._xyz
    after = render(cam, gm, pipe, bg)["render"].detach()
    d = (after - before).abs()
    assert float(d.mean()) < 2e-6 and float((d > 1e-4).float().mean()) < 2e-3 and float(d.max()) < 2e-2


def test_stage_three_rgb_reconstruction_step_descends():
    """system.StageThreeStep (GaussianIP.py:424-436): crop / half-size / L1 against refined images in refinement order;
    a few Adam steps on the colours reduce the loss, the gradient reaches every parameter group."""
    from argparse import ArgumentParser
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.guidance.refine import VIEW_IDX_ALL
    from gaussianip_amd.renderer import render_views
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.system import StageThreeStep
    from gaussianip_amd.utils import BasicPointCloud
    rng = np.random.default_rng(5)
    P = 20000
    pts = scenes.human_points(P, rng).astype(np.float32)
    gm = GaussianModel(0)
    gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    pipe = PipelineParams(ArgumentParser())
    bg = torch.zeros(3, device="cuda")
    H = W = 1024
    cams = [_camera(5.0, -180.0 + 11.25 * i, 1.8, 70.0, H, W) for i in range(32)]      # the 32-view refine orbit
    with torch.no_grad():
        target = torch.cat([render_views(cams[i:i + 8], gm, pipe, bg)["render"] for i in range(0, 32, 8)])   # orbit order
        target = (target * 0.5 + 0.25).permute(0, 2, 3, 1)                               # "refined": recoloured renders
    refined = target[torch.as_tensor(VIEW_IDX_ALL, device="cuda")]                       # refinement order, like refine_rgb returns
    st3 = StageThreeStep(gm, pipe, bg, cams, refined, VIEW_IDX_ALL, train_bs=4)
    assert st3.orbit_ids == list(range(32)) and st3.gt_small.shape == (32, 3, 415, 290)
    ids = [0, 9, 17, 30]
    losses = []
    for _ in range(6):
        out = st3.training_step(id_list=ids)
        gm.optimizer.zero_grad(set_to_none=True)
        out["loss"].backward()
        if not losses:
            for grp in gm.optimizer.param_groups:
                gr = grp["params"][0].grad
                assert gr is not None and torch.isfinite(gr).all(), grp["name"]
            assert float(gm._features_dc.grad.abs().max()) > 0 and float(gm._xyz.grad.abs().max()) > 0
        gm.optimizer.step()
        losses.append(float(out["loss"]))
    assert losses[-1] < losses[0], losses


def test_stage_three_step_with_lpips_term():
    """StageThreeStep with lambda_l1 = 10, lambda_lpips = 15 (configs/exp.yaml:125-126) and the LPIPS-VGG restatement
    (guidance/perceptual.py): target features cached once in fp16, the loss exceeds the L1-only loss, its gradient
    reaches the Gaussians through the MFMA convolution's data-gradient path."""
    from argparse import ArgumentParser
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.guidance.perceptual import LPIPSVGG
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.system import StageThreeStep
    from gaussianip_amd.utils import BasicPointCloud
    rng = np.random.default_rng(6)
    P = 20000
    pts = scenes.human_points(P, rng).astype(np.float32)
    gm = GaussianModel(0)
    gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    pipe = PipelineParams(ArgumentParser())
    bg = torch.ones(3, device="cuda")
    cams = [_camera(17.0, -180.0 + 90.0 * i, 1.5, 70.0, 1024, 1024) for i in range(4)]
    g = torch.Generator(device="cuda").manual_seed(1)
    refined = torch.rand(4, 1024, 1024, 3, device="cuda", generator=g)
    lp = LPIPSVGG().init_for_benchmark(2).prepare_inference("cuda")
    order = [2, 0, 3, 1]
    st3 = StageThreeStep(gm, pipe, bg, cams, refined, order, lambda_l1=10.0, lambda_lpips=15.0, perceptual=lp, train_bs=2)
    assert st3.gt_feats is not None and st3.gt_feats[0].shape == (4, 64, 415, 290) and st3.gt_feats[0].dtype == torch.float16
    out = st3.training_step(id_list=[1, 3])
    plain = StageThreeStep(gm, pipe, bg, cams, refined, order, lambda_l1=10.0, lambda_lpips=0.0, train_bs=2).training_step(id_list=[1, 3])
    assert float(out["loss"]) > float(plain["loss"]) > 0
    gm.optimizer.zero_grad(set_to_none=True)
    out["loss"].backward()
    assert float(gm._features_dc.grad.abs().max()) > 0 and torch.isfinite(gm._xyz.grad).all()


def test_exchange_bucket_pack_and_unpack_kernels():
    """gip_pack_bucket / gip_unpack_bucket (include/gip_model.h; the per-step multi-GPU exchange of parallel.py) against
    torch.cat / vector_norm / split on odd sizes and unaligned segment starts."""
    import ctypes
    from gaussianip_amd import _lib
    lib = _lib.model_lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    P, V = 1237, 3
    shapes = [(P, 3), (P, 1, 3), (P, 1), (P, 3), (P, 4), (5,)]
    segs = [torch.randn(sh, device="cuda", generator=g) for sh in shapes]
    g2d = torch.randn(V, P, 3, device="cuda", generator=g)
    counts = [t.numel() for t in segs]
    flat = torch.full((sum(counts) + P,), float("nan"), device="cuda")
    ptrs = (ctypes.c_void_p * len(segs))(*[t.data_ptr() for t in segs])
    cnt = (ctypes.c_int64 * len(segs))(*counts)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.gip_pack_bucket(ptrs, cnt, len(segs), ctypes.c_void_p(g2d.data_ptr()), V, P, ctypes.c_void_p(flat.data_ptr()), stream) == 0
    want = torch.cat([t.reshape(-1) for t in segs] + [torch.linalg.vector_norm(g2d[..., :2], dim=-1).sum(0)])
    assert torch.equal(flat[:sum(counts)], want[:sum(counts)])
    assert torch.allclose(flat[sum(counts):], want[sum(counts):], rtol=1e-6, atol=1e-7)
    # unpack: the first five segments scaled by 1/4, the sixth (a ready-made statistic) and nothing else unscaled
    outs = [torch.zeros_like(t) for t in segs]
    optr = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    assert lib.gip_unpack_bucket(optr, cnt, 5, ctypes.c_void_p(outs[5].data_ptr()), counts[5], ctypes.c_void_p(flat.data_ptr()), 0.25, stream) == 0
    for i in range(5):
        assert torch.equal(outs[i], segs[i] * 0.25)
    assert torch.equal(outs[5], segs[5])
    # no tail, no scaling
    outs2 = [torch.zeros_like(t) for t in segs]
    optr2 = (ctypes.c_void_p * len(outs2))(*[t.data_ptr() for t in outs2])
    assert lib.gip_unpack_bucket(optr2, cnt, len(segs), ctypes.c_void_p(None), 0, ctypes.c_void_p(flat.data_ptr()), 1.0, stream) == 0
    assert all(torch.equal(a, b) for a, b in zip(outs2, segs))
    # MAX bucket: radii maximum over the views + depth maximum as its bit pattern
    radii = torch.randint(0, 300, (V, P), device="cuda", generator=g, dtype=torch.int32)
    depth = torch.rand(V, 1, 37, 53, device="cuda", generator=g) * 7.0
    mb = torch.full((P + 1,), -5, device="cuda", dtype=torch.int32)
    assert lib.gip_max_bucket(ctypes.c_void_p(radii.data_ptr()), V, P, ctypes.c_void_p(depth.data_ptr()), depth.numel(),
                              ctypes.c_void_p(mb.data_ptr()), stream) == 0
    assert torch.equal(mb[:P], radii.amax(0)) and float(mb[P:].view(torch.float32)) == float(depth.max())
    from gaussianip_amd import parallel
    r2, d2 = parallel.exchange_forward_stats(radii, depth).wait()            # single process: the same numbers, no collective
    assert torch.equal(r2, radii.amax(0)) and float(d2) == float(depth.max())



def test_the_binding_printed_in_integration_md_works():
    """INTEGRATION.md §3 shows the ctypes binding a maintainer of the reference would write in place of
    `_C.rasterize_gaussians`: the printed code is executed as it stands (only the library path is made absolute) and must
    give the shipped wrapper's images."""
    import os
    import re
    from gaussianip_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def rasterize_gaussians" in b]
    assert len(blocks) == 1
    code = blocks[0].replace('ctypes.CDLL("libgip_raster.so")', 'ctypes.CDLL(%r)' % os.path.join(_lib.LIB_DIR, "libgip_raster.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    P, H, W = 3000, 96, 128
    sc = scenes.make_scene("stress", P, seed=4, sh_degree=0)
    cam = scenes.camera(10.0, 20.0, 1.6, 60.0, H, W)
    dev = "cuda"
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
    vm, pm, cp = (torch.from_numpy(cam[k]).to(dev) for k in ("viewmatrix", "projmatrix", "campos"))
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    color, radii, depth, alpha, state = ns["rasterize_gaussians"](
        bg, t["means3D"], None, t["opacities"], t["scales"], t["rotations"], 1.0, None, vm, pm, cam["tanfovx"], cam["tanfovy"], H, W,
        t["shs"], 0, cp, False, False)
    torch.cuda.synchronize()
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg,
                                       scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=0, campos=cp, prefiltered=False,
                                       debug=False)
    with torch.no_grad():
        c2, r2, d2, a2 = GaussianRasterizer(st)(means3D=t["means3D"], means2D=None, opacities=t["opacities"], shs=t["shs"],
                                                 scales=t["scales"], rotations=t["rotations"])
    assert torch.equal(color, c2) and torch.equal(radii, r2) and torch.equal(depth, d2) and torch.equal(alpha, a2)
    assert int(state[:64].view(torch.int32)[0]) == _lib.raster_lib().gip_abi_version()


def test_single_launch_adam_matches_torch_adam():
    """scene/adam.py::GipAdam (gip_adam_step: all parameter groups in one launch) against torch.optim.Adam on the same
    gradients: parameters, both moments and the step counts after six steps with per-group learning rates, a learning-rate
    change in between (update_learning_rate) and one GradScaler-skipped step (found_inf)."""
    from gaussianip_amd.scene.adam import GipAdam
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes, lrs = [(5000, 3), (5000, 1, 3), (5000, 0, 3), (5000, 1), (5000, 4)], [1.6e-4, 0.0125, 6e-4, 0.01, 0.001]
    ref_p = [torch.nn.Parameter(torch.randn(s, device="cuda", generator=g)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    mk = lambda ps: [{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))]      # noqa: E731
    ref, ours = torch.optim.Adam(mk(ref_p), lr=0.0, eps=1e-15), GipAdam(mk(our_p), lr=0.0, eps=1e-15)
    for it in range(6):
        grads = [torch.randn(s, device="cuda", generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,))) for s in shapes]
        for p, q, gr in zip(ref_p, our_p, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 3:
            ref.param_groups[0]["lr"] = ours.param_groups[0]["lr"] = 1.1e-4
        if it == 4:           # a skipped step: torch.amp hands found_inf to an optimizer that supports it; torch's plain Adam simply is not called
            ours.found_inf, ours.grad_scale = torch.ones(1, device="cuda"), None
            ours.step()
            del ours.found_inf, ours.grad_scale
            continue
        ref.step()
        ours.step()
    for p, q in zip(ref_p, our_p):
        if p.numel() == 0:
            continue
        assert float((p - q).abs().max()) <= 2e-6 * float(p.abs().max()), (tuple(p.shape), float((p - q).abs().max()), float(p.abs().max()))
        sr, so = ref.state[p], ours.state[q]
        assert float(so["step"]) == float(sr["step"]) == 5.0
        for key in ("exp_avg", "exp_avg_sq"):
            assert float((sr[key] - so[key]).abs().max()) <= 1e-6 * float(sr[key].abs().max()) + 1e-30
    # and it is what GaussianModel.training_setup(fused=True) builds
    from argparse import ArgumentParser
    from gaussianip_amd.arguments import OptimizationParams
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.utils import BasicPointCloud
    gm = GaussianModel(0)
    pts = np.random.default_rng(0).normal(size=(500, 3)).astype(np.float32)
    gm.create_from_pcd(BasicPointCloud(pts, np.full((500, 3), 0.5, np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()), fused=True)
    assert isinstance(gm.optimizer, GipAdam) and [g_["name"] for g_ in gm.optimizer.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
