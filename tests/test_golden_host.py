"""Host-side mirrors and the CPU oracle against golden vectors captured from the IMPORTED reference modules
(tools/make_golden.py).  Runs on CPU."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def g(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_projection_matrix_and_fov_helpers():
    from gaussianip_amd.utils import focal2fov, fov2focal, getProjectionMatrix
    d = g("projection.npz")
    for (fx, fy), P, (foc, fov) in zip(d["fovs"], d["P"], d["fov2focal_focal2fov"]):
        np.testing.assert_array_equal(getProjectionMatrix(float(d["znear"]), float(d["zfar"]), fx, fy).numpy(), P)
        assert fov2focal(fx, 1024) == foc
        assert focal2fov(fov2focal(fy, 512), 1024) == fov


def test_eval_sh_and_colour_offset():
    from gaussianip_amd.utils import RGB2SH, SH2RGB, eval_sh
    d = g("eval_sh.npz")
    sh, dirs = torch.from_numpy(d["sh"]), torch.from_numpy(d["dirs"])
    for deg in range(4):
        np.testing.assert_allclose(eval_sh(deg, sh, dirs).numpy(), d["deg%d" % deg], rtol=0, atol=2e-6)
    rgb = torch.from_numpy(d["rgb"])
    np.testing.assert_allclose(RGB2SH(rgb).numpy(), d["rgb2sh"], atol=1e-6)
    np.testing.assert_allclose(SH2RGB(rgb).numpy(), d["sh2rgb"], atol=1e-6)


def test_covariance_and_rotation():
    from gaussianip_amd.utils import build_rotation, build_scaling_rotation, inverse_sigmoid, strip_symmetric
    d = g("covariance.npz")
    s, q = torch.from_numpy(d["scales"]), torch.from_numpy(d["rotations"])
    np.testing.assert_allclose(build_rotation(q).numpy(), d["R"], atol=1e-6)
    L = build_scaling_rotation(float(d["scale_modifier"]) * s, q)
    np.testing.assert_allclose(strip_symmetric(L @ L.transpose(1, 2)).numpy(), d["cov6"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(inverse_sigmoid(torch.from_numpy(d["inv_sigmoid_x"])).numpy(), d["inv_sigmoid_y"], atol=1e-6)


def test_lr_schedule_and_argument_defaults():
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.utils import get_expon_lr_func
    from argparse import ArgumentParser
    d = g("lr_schedule.npz")
    op, pp = OptimizationParams(ArgumentParser()), PipelineParams(ArgumentParser())
    np.testing.assert_array_equal(np.array([op.position_lr_init, op.position_lr_final, op.position_lr_delay_mult,
                                            op.position_lr_max_steps, op.feature_lr, op.opacity_lr, op.scaling_lr,
                                            op.rotation_lr, op.percent_dense]), d["opt"])
    np.testing.assert_array_equal(np.array([int(pp.convert_SHs_python), int(pp.compute_cov3D_python), int(pp.debug)]), d["pipe"])
    f = get_expon_lr_func(lr_init=op.position_lr_init * 4.0, lr_final=op.position_lr_final * 4.0,
                          lr_delay_mult=op.position_lr_delay_mult, max_steps=op.position_lr_max_steps)
    np.testing.assert_allclose([f(int(t)) for t in d["steps"]], d["lr"], rtol=1e-12)
    f2 = get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=1000)
    np.testing.assert_allclose([f2(int(t)) for t in d["steps"]], d["lr_delay"], rtol=1e-12)
    assert f(-1) == 0.0


def test_camera_matrices():
    from gaussianip_amd.scene import Camera
    d = g("cameras.npz")
    for inp, out in zip(d["inputs"], d["outputs"]):
        c2w = torch.from_numpy(inp[:16].reshape(4, 4)).float()
        cam = Camera(c2w=c2w, FoVy=float(inp[16]), height=int(inp[17]), width=int(inp[18]), data_device="cpu")
        np.testing.assert_allclose(cam.world_view_transform.numpy().reshape(-1), out[0:16], atol=2e-6)
        np.testing.assert_allclose(cam.projection_matrix.numpy().reshape(-1), out[16:32], atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy().reshape(-1), out[32:48], atol=5e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), out[48:51], atol=5e-6)
        assert abs(cam.FoVx - out[51]) < 1e-12 and abs(cam.FoVy - out[52]) < 1e-7
        assert (cam.znear, cam.zfar) == (out[53].astype(np.float32), out[54])


def _model_from_golden(d, prefix):
    """GaussianModel on CPU holding the golden `pre_*` state, with Adam moments installed."""
    from argparse import ArgumentParser
    from gaussianip_amd.arguments import OptimizationParams
    from gaussianip_amd.scene import GaussianModel
    from torch import nn
    gm = GaussianModel(0, device="cpu")
    gm.spatial_lr_scale = float(d["spatial_lr_scale"])
    t = lambda k: nn.Parameter(torch.from_numpy(d[prefix + k].copy()).requires_grad_(True))  # noqa: E731
    gm._xyz, gm._features_dc, gm._features_rest = t("xyz"), t("f_dc"), t("f_rest")
    gm._scaling, gm._rotation, gm._opacity = t("scaling"), t("rotation"), t("opacity")
    gm.max_radii2D = torch.zeros(gm._xyz.shape[0])
    gm.training_setup(OptimizationParams(ArgumentParser()))
    for grp in gm.optimizer.param_groups:
        gm.optimizer.state[grp["params"][0]] = dict(step=torch.tensor(3.0),
                                                    exp_avg=torch.from_numpy(d["pre_m_" + grp["name"]].copy()),
                                                    exp_avg_sq=torch.from_numpy(d["pre_v_" + grp["name"]].copy()))
    return gm


def test_gaussian_model_init_getters_and_optimizer_groups():
    from argparse import ArgumentParser
    from gaussianip_amd.arguments import OptimizationParams
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.utils import BasicPointCloud
    d = g("gaussian_model.npz")
    pts = d["points"]
    d2 = ((pts[:, None, :].astype(np.float64) - pts[None, :, :].astype(np.float64)) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    dist2 = np.sort(d2, axis=1)[:, :3].mean(1).astype(np.float32)   # what distCUDA2 returns for these points
    gm = GaussianModel(0, device="cpu")
    gm.create_from_pcd(BasicPointCloud(pts, d["colors"], None), 4.0, dist2=dist2)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    for k, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("scaling", "_scaling"),
                    ("rotation", "_rotation"), ("opacity", "_opacity")):
        np.testing.assert_allclose(getattr(gm, attr).detach().numpy(), d["init_" + k], rtol=2e-6, atol=1e-6)
    for k in ("get_scaling", "get_opacity", "get_rotation", "get_features"):
        np.testing.assert_allclose(getattr(gm, k).detach().numpy(), d["getter_" + k], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(gm.get_covariance(1.0).detach().numpy(), d["getter_get_covariance"], rtol=1e-5, atol=1e-10)
    assert [grp["name"] for grp in gm.optimizer.param_groups] == list(d["names"])
    np.testing.assert_allclose([grp["lr"] for grp in gm.optimizer.param_groups], d["lrs"], rtol=1e-12)
    assert gm.optimizer.defaults["eps"] == 1e-15


def test_densify_and_prune_then_prune_only_match_reference_trace():
    d = g("gaussian_model.npz")
    gm = _model_from_golden(d, "pre_")
    gm.xyz_gradient_accum = torch.from_numpy(d["pre_grad_accum"].copy())
    gm.denom = torch.from_numpy(d["pre_denom"].copy())
    torch.manual_seed(int(d["seed_before_densify"]))
    gm.densify_and_prune(0.0002, 0.05, 4.0, None, 0.015)
    assert gm._xyz.shape[0] == d["post_xyz"].shape[0]
    for k, attr in (("xyz", "_xyz"), ("scaling", "_scaling"), ("rotation", "_rotation"), ("opacity", "_opacity"),
                    ("f_dc", "_features_dc")):
        np.testing.assert_allclose(getattr(gm, attr).detach().numpy(), d["post_" + k], rtol=1e-6, atol=1e-7)
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state[grp["params"][0]]
        np.testing.assert_array_equal(st["exp_avg"].numpy(), d["post_m_" + grp["name"]])
        np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), d["post_v_" + grp["name"]])
        assert grp["params"][0] is getattr(gm, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest",
                                                "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}[grp["name"]])
    np.testing.assert_array_equal(gm.max_radii2D.numpy(), d["post_max_radii2D"])
    np.testing.assert_array_equal(gm.denom.numpy(), d["post_denom"])
    gm.prune_only(min_opacity=0.05, max_world_size=0.01)
    np.testing.assert_allclose(gm._xyz.detach().numpy(), d["post2_xyz"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gm._opacity.detach().numpy(), d["post2_opacity"], rtol=1e-6, atol=1e-7)


def test_add_densification_stats_matches_reference():
    d = g("gaussian_model.npz")
    gm = _model_from_golden(d, "pre_")
    vs, vis = torch.from_numpy(d["pre_viewspace"]), torch.from_numpy(d["pre_vis"])
    gm.add_densification_stats(vs, vis)
    gm.add_densification_stats(vs * 0.5, vis)
    np.testing.assert_allclose(gm.xyz_gradient_accum.numpy(), d["pre_grad_accum"], rtol=1e-6)
    np.testing.assert_array_equal(gm.denom.numpy(), d["pre_denom"])


def test_ply_round_trip(tmp_path):
    d = g("gaussian_model.npz")
    gm = _model_from_golden(d, "pre_")
    path = str(tmp_path / "pc" / "it.ply")
    gm.save_ply(path)
    head = open(path, "rb").read(400).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n" % gm._xyz.shape[0])
    assert "property float f_dc_2\nproperty float opacity\nproperty float scale_0" in head
    from gaussianip_amd.scene import GaussianModel
    g2 = GaussianModel(0, device="cpu")
    g2.load_ply(path)
    for attr in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(g2, attr).detach(), getattr(gm, attr).detach()), attr


# ---------------------------------------------------------------------------------------------------------
# the ORACLE's pre-stages against the same reference vectors (this is what "pins" the oracle, see its header)
# ---------------------------------------------------------------------------------------------------------
def _view_all(n_points_center, H=64, W=64):
    from dense_reference import look_at_camera
    view, proj, campos, tanx, tany = look_at_camera(0.0, 0.0, 6.0, 40.0, H, W)
    return view.numpy().astype(np.float32), proj.numpy().astype(np.float32), tanx, tany


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_colour_matches_reference_eval_sh(oracle, deg):
    d = g("eval_sh.npz")
    sh, dirs = d["sh"], d["dirs"]                         # [P,3,16], unit dirs
    P = sh.shape[0]
    view, proj, tanx, tany = _view_all(P)
    c0 = np.array([0.1, -0.2, 0.05], np.float32)          # SH direction origin (need not be the view origin)
    pos = (c0[None, :] + dirs).astype(np.float32)
    shs = np.ascontiguousarray(np.transpose(sh, (0, 2, 1)))   # -> [P,16,3] rasterizer layout
    ro = oracle.RasterOracle()
    ro.forward(image_height=64, image_width=64, tanfovx=tanx, tanfovy=tany, bg=np.zeros(3, np.float32), scale_modifier=1.0,
               viewmatrix=view, projmatrix=proj, sh_degree=deg, campos=c0, means3D=pos,
               opacities=np.full((P, 1), 0.5, np.float32), shs=shs, scales=np.full((P, 3), 0.01, np.float32),
               rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)))
    # direction actually used by the rasterizer: normalize(pos - campos) in float32
    dd = pos - c0[None, :]
    dd = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    from gaussianip_amd.utils import eval_sh as our_eval
    ref = np.maximum(our_eval(deg, torch.from_numpy(sh), torch.from_numpy(dd.astype(np.float32))).numpy() + 0.5, 0.0)
    np.testing.assert_allclose(ro.geom()["rgb"], ref, atol=3e-6)
    # and against the reference's own numbers on the golden directions (dirs differ from dd only by float rounding)
    np.testing.assert_allclose(ro.geom()["rgb"], d["rgb%d" % deg], atol=2e-5)


def test_oracle_cov3D_matches_reference_covariance(oracle):
    d = g("covariance.npz")
    s, q = d["scales"], d["rotations"]
    qn = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)   # the Python side normalises (gaussian_model.py:88-89)
    P = s.shape[0]
    view, proj, tanx, tany = _view_all(P)
    rng = np.random.default_rng(0)
    pos = rng.uniform(-0.5, 0.5, (P, 3)).astype(np.float32)
    ro = oracle.RasterOracle()
    ro.forward(image_height=64, image_width=64, tanfovx=tanx, tanfovy=tany, bg=np.zeros(3, np.float32),
               scale_modifier=float(d["scale_modifier"]), viewmatrix=view, projmatrix=proj, sh_degree=0,
               campos=np.zeros(3, np.float32), means3D=pos, opacities=np.full((P, 1), 0.5, np.float32),
               shs=np.zeros((P, 1, 3), np.float32), scales=s, rotations=qn)
    np.testing.assert_allclose(ro.geom()["cov3D"], d["cov6"], rtol=2e-5, atol=1e-9)


def test_oracle_projection_matches_reference_camera(oracle):
    """means2D / depth from the oracle vs the reference Camera matrices applied in float64."""
    d = g("cameras.npz")
    inp, out = d["inputs"][40], d["outputs"][40]
    H, W = int(inp[17]), int(inp[18])
    V = out[0:16].reshape(4, 4)
    VP = out[32:48].reshape(4, 4)
    rng = np.random.default_rng(1)
    pos = rng.uniform(-0.4, 0.4, (200, 3)).astype(np.float32)
    ro = oracle.RasterOracle()
    ro.forward(image_height=H, image_width=W, tanfovx=math.tan(out[51] / 2), tanfovy=math.tan(out[52] / 2),
               bg=np.zeros(3, np.float32), scale_modifier=1.0, viewmatrix=V.astype(np.float32),
               projmatrix=VP.astype(np.float32), sh_degree=0, campos=out[48:51].astype(np.float32), means3D=pos,
               opacities=np.full((200, 1), 0.5, np.float32), shs=np.zeros((200, 1, 3), np.float32),
               scales=np.full((200, 3), 0.01, np.float32), rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (200, 1)))
    ph = np.concatenate([pos.astype(np.float64), np.ones((200, 1))], 1)
    hom = ph @ VP
    ndc = hom[:, :2] / (hom[:, 3:4] + 1e-7)
    pix = np.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], 1)
    zview = (ph @ V)[:, 2]
    geo = ro.geom()
    np.testing.assert_allclose(geo["means2D"], pix, atol=2e-3)
    np.testing.assert_allclose(geo["depths"], zview, atol=1e-5)


def test_knn_oracle_matches_golden(oracle):
    d = g("knn_dist2.npz")
    np.testing.assert_allclose(oracle.knn_mean_dist2(d["points"]), d["dist2"], rtol=2e-5, atol=1e-9)


def test_c_abi_library_exports_every_declared_symbol():
    import re
    from gaussianip_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _lib.raster_lib()
    decl = re.findall(r"\b(gip_[a-z_0-9]+)\s*\(", open(os.path.join(root, "include", "gip_raster.h")).read())
    assert set(decl) >= {"gip_raster_forward", "gip_raster_backward", "gip_raster_state_bytes"}
    for sym in set(decl):
        assert hasattr(lib, sym), sym
    assert lib.gip_abi_version() == 4
    cfg = _lib.GipRasterConfig()
    cfg.P, cfg.V, cfg.H, cfg.W, cfg.sh_degree, cfg.sh_coeffs, cfg.capacity = 1000, 2, 64, 48, 0, 1, 1 << 16
    import ctypes
    assert lib.gip_raster_state_bytes(ctypes.byref(cfg)) > 0 and lib.gip_raster_scratch_bytes(ctypes.byref(cfg)) == (1 << 16) * 48
    cfg.V = 99
    assert lib.gip_raster_state_bytes(ctypes.byref(cfg)) == 0
    for header, loader in (("gip_knn.h", _lib.knn_lib), ("gip_nn.h", _lib.nn_lib), ("gip_model.h", _lib.model_lib), ("gip_pose.h", _lib.model_lib)):
        other = loader()
        for sym in set(re.findall(r"\b(gip_[a-z_0-9]+)\s*\(", open(os.path.join(root, "include", header)).read())):
            assert hasattr(other, sym), (header, sym)


def test_refine_orbit_and_system_config_from_yaml_keys():
    """create_refine_batch (GaussianIP.py:232-281) geometry and StageOneConfig.from_dict on the reference's system keys."""
    from gaussianip_amd.system import StageOneConfig, create_refine_batch
    b = create_refine_batch()
    c2w = b["c2w"]
    assert c2w.shape == (32, 4, 4) and b["azimuth"][0] == -180.0 and abs(float(b["azimuth"][1] - b["azimuth"][0]) - 11.25) < 1e-5
    R = c2w[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(32, 3, 3), atol=1e-5)
    pos = c2w[:, :3, 3]
    assert torch.allclose(pos.norm(dim=-1), torch.full((32,), 1.5), atol=1e-5)
    assert torch.allclose(pos[:, 2], torch.full((32,), 1.5 * math.sin(math.radians(17.0))), atol=1e-5)
    assert torch.allclose(F.normalize(-pos, dim=-1), -R[:, :, 2], atol=1e-5)            # the camera looks at the origin (-z forward)
    assert torch.allclose(b["fovy"], torch.full((32,), math.radians(70.0)))
    cfg = StageOneConfig.from_dict({"densify_prune_start_step": 100, "max_grad": 3e-4, "refine_n_views": 16, "pts_num": 100000,
                                    "loss": {"lambda_sds": 2.0, "lambda_sparsity": 0.5, "lambda_opaque": 0, "scale_tau": 2}, "stage": "stage1"})
    assert cfg.densify_prune_start_step == 100 and cfg.max_grad == 3e-4 and cfg.lambda_sds == 2.0 and cfg.lambda_sparsity == 0.5
    assert cfg.refine_n_views == 16 and cfg.extra["pts_num"] == 100000 and "loss" in cfg.extra


def test_ply_file_matches_the_reference_vertex_table(tmp_path):
    """save_ply writes exactly the vertex table the reference hands to plyfile (gaussian_model.py:199-216; fixture
    tools/make_golden.py `ply`): same property names in the same order, same float32 rows, in plyfile's
    binary_little_endian layout; load_ply restores the parameters (and reads files whose columns come in any order)."""
    from gaussianip_amd.scene import GaussianModel
    d = np.load(os.path.join(GOLD, "ply_layout.npz"))
    assert str(d["element_name"]) == "vertex"
    for deg in (0, 2):
        gm = GaussianModel(deg)
        gm.device = torch.device("cpu")
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            setattr(gm, k, torch.nn.Parameter(torch.from_numpy(d["deg%d%s" % (deg, k)])))
        names = list(d["names_deg%d" % deg])
        assert gm.construct_list_of_attributes() == names
        path = str(tmp_path / ("m%d.ply" % deg))
        gm.save_ply(path)
        raw = open(path, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        lines = head.decode("ascii").split("\n")
        n = d["table_deg%d" % deg].shape[0]
        assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex %d" % n]
        assert lines[3:-1] == ["property float %s" % k for k in names] and lines[-1] == ""
        table = np.frombuffer(body, dtype="<f4").reshape(n, len(names))
        assert np.array_equal(table, d["table_deg%d" % deg])
        back = GaussianModel(deg)
        back.device = torch.device("cpu")
        back.load_ply(path)
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            assert torch.equal(getattr(back, k).detach(), torch.from_numpy(d["deg%d%s" % (deg, k)])), k


def test_ctypes_structures_match_the_c_headers(tmp_path):
    """sizeof / offsetof of every C-ABI structure, taken from include/gip_raster.h by the C compiler, equal what the ctypes
    mirror in gaussianip_amd/_lib.py declares (a field added on one side only would shift every later argument silently)."""
    import ctypes
    import subprocess
    from gaussianip_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"GipRasterConfig": _lib.GipRasterConfig, "GipRasterInputs": _lib.GipRasterInputs, "GipRasterOutputs": _lib.GipRasterOutputs,
               "GipRasterGradsIn": _lib.GipRasterGradsIn, "GipRasterGradsOut": _lib.GipRasterGradsOut,
               "GipRasterStateLayout": _lib.GipRasterStateLayout}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gip_raster.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, fname, name, fname))
    lines.append('  printf("GipRasterHeader %zu\\n", sizeof(GipRasterHeader));')
    lines.append('  printf("GIP_MAX_VIEWS %d\\nGIP_PARTIAL_FLOATS %d\\nGIP_ABI_VERSION %d\\n", GIP_MAX_VIEWS, GIP_PARTIAL_FLOATS, GIP_ABI_VERSION);')
    lines += ['  return 0;', '}']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (name, fname)]) == getattr(cls, fname).offset, (name, fname)
    assert int(got["GipRasterHeader"]) == 64
    assert int(got["GIP_MAX_VIEWS"]) == _lib.GIP_MAX_VIEWS and int(got["GIP_PARTIAL_FLOATS"]) == _lib.GIP_PARTIAL_FLOATS
    assert int(got["GIP_ABI_VERSION"]) == _lib.raster_lib().gip_abi_version()
