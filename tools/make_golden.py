#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference's own Python modules from /root/reference
(read-only, this container only).  Only arrays are written: inputs and the reference's outputs.  No reference
source, bytecode or stub travels with the repo.  Recipe: SURVEY.md Appendix B.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
os.makedirs(OUT, exist_ok=True)


# ---- device shim: the reference hard-codes device="cuda" / .cuda() (general_utils.py:65,83,102; cameras.py:48-49) ----
def _install_device_shim():
    def wrap(fn):
        def inner(*a, **k):
            d = k.get("device", None)
            if d is not None and str(d).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    for name in ("zeros", "ones", "empty", "tensor", "zeros_like", "ones_like", "full", "rand", "randn", "normal", "arange"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None


def _install_stub_modules():
    ply = types.ModuleType("plyfile")
    ply.PlyData = object
    ply.PlyElement = object
    sys.modules["plyfile"] = ply
    knn = types.ModuleType("simple_knn")
    knn_c = types.ModuleType("simple_knn._C")

    def distCUDA2(pts):
        d = torch.cdist(pts.double(), pts.double()) ** 2
        d.fill_diagonal_(float("inf"))
        return d.topk(3, largest=False).values.mean(1).float()
    knn_c.distCUDA2 = distCUDA2
    knn._C = knn_c
    sys.modules["simple_knn"] = knn
    sys.modules["simple_knn._C"] = knn_c
    dgr = types.ModuleType("diff_gaussian_rasterization")
    dgr.GaussianRasterizationSettings = object
    dgr.GaussianRasterizer = object
    sys.modules["diff_gaussian_rasterization"] = dgr


def orbit_c2w(elev_deg, azim_deg, dist):
    """threestudio camera-to-world (camera_data.py:423-454): z up, camera looks at the origin."""
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    pos = torch.tensor([dist * math.cos(el) * math.cos(az), dist * math.cos(el) * math.sin(az), dist * math.sin(el)])
    center = torch.zeros(3)
    up = torch.tensor([0.0, 0.0, 1.0])
    lookat = torch.nn.functional.normalize(center - pos, dim=-1)
    right = torch.nn.functional.normalize(torch.linalg.cross(lookat, up), dim=-1)
    upv = torch.nn.functional.normalize(torch.linalg.cross(right, lookat), dim=-1)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, upv, -lookat, pos
    return c2w


def main():
    _install_device_shim()
    _install_stub_modules()
    from gaussiansplatting.utils import sh_utils, graphics_utils, general_utils
    from gaussiansplatting.arguments import OptimizationParams, PipelineParams
    from argparse import ArgumentParser

    g = torch.Generator().manual_seed(42)

    # (7) projection matrix + fov helpers  (graphics_utils.py:73-99)
    fovs = [(0.7, 0.9), (1.2217, 1.2217), (0.6981, 1.0), (1.0, 0.5)]
    P = np.stack([graphics_utils.getProjectionMatrix(0.01, 100.0, fx, fy).numpy() for fx, fy in fovs])
    f2f = np.array([[graphics_utils.fov2focal(fx, 1024), graphics_utils.focal2fov(graphics_utils.fov2focal(fy, 512), 1024)] for fx, fy in fovs])
    np.savez(os.path.join(OUT, "projection.npz"), fovs=np.array(fovs), znear=0.01, zfar=100.0, P=P, fov2focal_focal2fov=f2f)

    # (2) eval_sh degrees 0..3 (+0.5 / clamp as gaussian_renderer/__init__.py:77-78)
    Pn = 256
    sh = torch.randn(Pn, 3, 16, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(Pn, 3, generator=g), dim=1)
    out = {}
    for deg in range(4):
        res = sh_utils.eval_sh(deg, sh, dirs)
        out["deg%d" % deg] = res.numpy()
        out["rgb%d" % deg] = torch.clamp_min(res + 0.5, 0.0).numpy()
    rgb = torch.rand(Pn, 3, generator=g)
    np.savez(os.path.join(OUT, "eval_sh.npz"), sh=sh.numpy(), dirs=dirs.numpy(), rgb=rgb.numpy(),
             rgb2sh=sh_utils.RGB2SH(rgb).numpy(), sh2rgb=sh_utils.SH2RGB(rgb).numpy(), C0=sh_utils.C0, **out)

    # (3) covariance from scaling / rotation (general_utils.py:64-110, gaussian_model.py:16-20)
    s = torch.rand(Pn, 3, generator=g) * 0.1 + 0.001
    q = torch.randn(Pn, 4, generator=g)  # NOT normalised: build_rotation normalises (general_utils.py:79-81)
    L = general_utils.build_scaling_rotation(1.7 * s, q)
    cov = general_utils.strip_symmetric(L @ L.transpose(1, 2))
    R = general_utils.build_rotation(q)
    x = torch.rand(64, generator=g) * 0.98 + 0.01
    np.savez(os.path.join(OUT, "covariance.npz"), scales=s.numpy(), rotations=q.numpy(), scale_modifier=1.7, cov6=cov.numpy(),
             R=R.numpy(), inv_sigmoid_x=x.numpy(), inv_sigmoid_y=general_utils.inverse_sigmoid(x).numpy())

    # (6) learning-rate schedule (general_utils.py:29-62 with the parameters of arguments/__init__.py:73-76 x spatial_lr_scale 4)
    op = OptimizationParams(ArgumentParser())
    pp = PipelineParams(ArgumentParser())
    f = general_utils.get_expon_lr_func(lr_init=op.position_lr_init * 4.0, lr_final=op.position_lr_final * 4.0,
                                        lr_delay_mult=op.position_lr_delay_mult, max_steps=op.position_lr_max_steps)
    steps = np.array([0, 1, 10, 100, 500, 1000, 2399, 2400, 10000, 30000, 40000])
    f2 = general_utils.get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=1000)
    np.savez(os.path.join(OUT, "lr_schedule.npz"), steps=steps, lr=np.array([f(int(t)) for t in steps]),
             lr_delay=np.array([f2(int(t)) for t in steps]),
             opt=np.array([op.position_lr_init, op.position_lr_final, op.position_lr_delay_mult, op.position_lr_max_steps,
                           op.feature_lr, op.opacity_lr, op.scaling_lr, op.rotation_lr, op.percent_dense]),
             pipe=np.array([int(pp.convert_SHs_python), int(pp.compute_cov3D_python), int(pp.debug)]))

    # (1) Camera matrices (cameras.py:17-51): the 36-view orbit (elev 5, dist 1.8, fovy 70) + 16 random training cameras
    from gaussiansplatting.scene.cameras import Camera
    cams_in, cams_out = [], []
    rng = np.random.default_rng(42)
    specs = [(5.0, 360.0 * i / 36, 1.8, 70.0, 1024, 1024) for i in range(36)]
    specs += [(rng.uniform(-30, 30), rng.uniform(-180, 180), rng.uniform(1.3, 1.7), rng.uniform(40, 70), 1024, 1024) for _ in range(12)]
    specs += [(rng.uniform(-30, 30), rng.uniform(-180, 180), rng.uniform(1.3, 1.7), rng.uniform(40, 70), 512, 768) for _ in range(4)]
    for el, az, dist, fovy_deg, H, W in specs:
        c2w = orbit_c2w(el, az, dist)
        fovy = math.radians(fovy_deg)
        cam = Camera(c2w=c2w.clone(), FoVy=fovy, height=H, width=W)
        cams_in.append(np.concatenate([c2w.numpy().reshape(-1), [fovy, H, W]]))
        cams_out.append(np.concatenate([cam.world_view_transform.numpy().reshape(-1), cam.projection_matrix.numpy().reshape(-1),
                                        cam.full_proj_transform.numpy().reshape(-1), cam.camera_center.numpy().reshape(-1),
                                        [cam.FoVx, cam.FoVy, cam.znear, cam.zfar]]))
    np.savez(os.path.join(OUT, "cameras.npz"), inputs=np.array(cams_in, dtype=np.float64), outputs=np.array(cams_out, dtype=np.float64),
             layout="inputs: c2w[16] fovy H W ; outputs: world_view[16] projection[16] full_proj[16] center[3] FoVx FoVy znear zfar")

    # (4) GaussianModel: create_from_pcd -> training_setup -> densify_and_prune / prune_only traces (gaussian_model.py:113-418)
    from gaussiansplatting.scene.gaussian_model import GaussianModel
    from gaussiansplatting.utils.graphics_utils import BasicPointCloud
    Pn = 600
    pts = (torch.rand(Pn, 3, generator=g) - 0.5).numpy().astype(np.float32)
    cols = torch.rand(Pn, 3, generator=g).numpy().astype(np.float32)
    gm = GaussianModel(0)
    gm.create_from_pcd(BasicPointCloud(pts, cols, None), 4.0)
    gm.training_setup(op)
    init = dict(xyz=gm._xyz.detach().numpy().copy(), f_dc=gm._features_dc.detach().numpy().copy(),
                f_rest=gm._features_rest.detach().numpy().copy(), scaling=gm._scaling.detach().numpy().copy(),
                rotation=gm._rotation.detach().numpy().copy(), opacity=gm._opacity.detach().numpy().copy())
    getters = dict(get_scaling=gm.get_scaling.detach().numpy().copy(), get_opacity=gm.get_opacity.detach().numpy().copy(),
                   get_rotation=gm.get_rotation.detach().numpy().copy(), get_features=gm.get_features.detach().numpy().copy(),
                   get_covariance=gm.get_covariance(1.0).detach().numpy().copy())
    # make the state interesting: anisotropic scales, varied opacity, a few Adam steps worth of moments
    with torch.no_grad():
        gm._scaling += torch.randn(Pn, 3, generator=g) * 0.8
        gm._opacity += torch.randn(Pn, 1, generator=g) * 2.0
        gm._rotation += torch.randn(Pn, 4, generator=g) * 0.3
    for grp in gm.optimizer.param_groups:
        p = grp["params"][0]
        gm.optimizer.state[p] = dict(step=torch.tensor(3.0), exp_avg=torch.randn(p.shape, generator=g) * 0.01,
                                     exp_avg_sq=torch.rand(p.shape, generator=g) * 0.001)
    vs = torch.randn(Pn, 3, generator=g) * 3e-4
    vis = torch.rand(Pn, generator=g) > 0.2
    gm.add_densification_stats(vs, vis)
    gm.add_densification_stats(vs * 0.5, vis)
    pre = dict(xyz=gm._xyz.detach().numpy().copy(), scaling=gm._scaling.detach().numpy().copy(),
               rotation=gm._rotation.detach().numpy().copy(), opacity=gm._opacity.detach().numpy().copy(),
               f_dc=gm._features_dc.detach().numpy().copy(), grad_accum=gm.xyz_gradient_accum.numpy().copy(),
               denom=gm.denom.numpy().copy(), viewspace=vs.numpy(), vis=vis.numpy(),
               exp_avg_xyz=gm.optimizer.state[gm.optimizer.param_groups[0]["params"][0]]["exp_avg"].numpy().copy(),
               f_rest=gm._features_rest.detach().numpy().copy())
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state[grp["params"][0]]
        pre["m_" + grp["name"]] = st["exp_avg"].numpy().copy()
        pre["v_" + grp["name"]] = st["exp_avg_sq"].numpy().copy()
    torch.manual_seed(1234)   # the split samples come from torch.normal on the global generator (gaussian_model.py:368)
    gm.densify_and_prune(0.0002, 0.05, 4.0, None, 0.015)
    post = dict(xyz=gm._xyz.detach().numpy().copy(), scaling=gm._scaling.detach().numpy().copy(),
                rotation=gm._rotation.detach().numpy().copy(), opacity=gm._opacity.detach().numpy().copy(),
                f_dc=gm._features_dc.detach().numpy().copy(),
                exp_avg_xyz=gm.optimizer.state[gm.optimizer.param_groups[0]["params"][0]]["exp_avg"].numpy().copy(),
                exp_avg_sq_scaling=gm.optimizer.state[gm.optimizer.param_groups[4]["params"][0]]["exp_avg_sq"].numpy().copy(),
                max_radii2D=gm.max_radii2D.numpy().copy(), denom=gm.denom.numpy().copy())
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state[grp["params"][0]]
        post["m_" + grp["name"]] = st["exp_avg"].numpy().copy()
        post["v_" + grp["name"]] = st["exp_avg_sq"].numpy().copy()
    gm.prune_only(min_opacity=0.05, max_world_size=0.01)
    post2 = dict(xyz=gm._xyz.detach().numpy().copy(), opacity=gm._opacity.detach().numpy().copy())
    np.savez(os.path.join(OUT, "gaussian_model.npz"), points=pts, colors=cols, spatial_lr_scale=4.0, seed_before_densify=1234,
             **{"init_" + k: v for k, v in init.items()}, **{"getter_" + k: v for k, v in getters.items()},
             **{"pre_" + k: v for k, v in pre.items()}, **{"post_" + k: v for k, v in post.items()},
             **{"post2_" + k: v for k, v in post2.items()},
             lrs=np.array([grp["lr"] for grp in gm.optimizer.param_groups]),
             names=np.array([grp["name"] for grp in gm.optimizer.param_groups]))

    # (8) distCUDA2 semantics on 4096 points — values of the brute-force STUB above (simple_knn.cu:147-183 semantics),
    #     NOT of the CUDA kernel, which cannot be built here.
    pts = (torch.rand(4096, 3, generator=g) - 0.5).float()
    np.savez(os.path.join(OUT, "knn_dist2.npz"), points=pts.numpy(), dist2=sys.modules["simple_knn._C"].distCUDA2(pts).numpy(),
             source="brute-force float64 cdist/topk stub (SURVEY.md Appendix B item 3), not the CUDA kernel")
    print("golden vectors written to", OUT)


def ahds():
    """(5) AHDS timestep table: loads threestudio/models/guidance/ipa_guidance.py by path with permissive stubs
     and calls the two schedule functions unbound."""
    class Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__") and name.endswith("__"):
                raise AttributeError(name)
            child = Stub(self.__name__ + "." + name)
            setattr(self, name, child)
            return child

        def __call__(self, *a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return self

        def __getitem__(self, k):
            return self

        def __mro_entries__(self, bases):
            return (object,)

    import typing
    names = ["threestudio", "threestudio.utils", "threestudio.utils.base", "threestudio.utils.misc", "threestudio.utils.typing",
             "threestudio.models", "threestudio.models.prompt_processors", "threestudio.models.prompt_processors.base",
             "threestudio.models.guidance", "threestudio.models.guidance.models", "threestudio.models.guidance.models.ip_adapter",
             "threestudio.models.guidance.models.ip_adapter.ip_adapter_faceid", "threestudio.models.guidance.models.pipeline_ipa",
             "threestudio.models.guidance.models.pipeline_ipa_controlnet", "diffusers", "diffusers.utils", "diffusers.utils.import_utils",
             "cv2", "insightface", "insightface.app", "insightface.utils", "PIL", "PIL.Image", "tqdm"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = Stub(n)
    ty = sys.modules["threestudio.utils.typing"]
    for k in dir(typing):
        if not k.startswith("_"):
            setattr(ty, k, getattr(typing, k))
    ty.Tensor = torch.Tensor
    jt = ["Bool", "Complex", "Float", "Inexact", "Int", "Integer", "Num", "Shaped", "UInt", "DictConfig", "typechecker"]
    for k in jt:
        setattr(ty, k, Stub("jaxtyping_stub"))
    ty.__all__ = [k for k in dir(typing) if not k.startswith("_")] + ["Tensor"] + jt

    class BaseObject:
        class Config:
            pass
    sys.modules["threestudio.utils.base"].BaseObject = BaseObject
    path = os.path.join(REF, "threestudio/models/guidance/ipa_guidance.py")
    spec = importlib.util.spec_from_file_location("threestudio.models.guidance.ipa_guidance", path)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "threestudio.models.guidance"
    spec.loader.exec_module(mod)
    G = mod.StableDiffusionGuidance
    self = object.__new__(G)
    # ipa_guidance.py:200-210 (1-D x0: the trailing comma at :206 makes a 2-D x0 that scipy >= 1.11 rejects)
    W = G.get_optimized_dual_gaussian(self, [260, 60, 280], [0.41, 0.21, 0.375], [(0, 350), (350, 450), (450, 800)], 800,
                                      [(200, 400), (20, 100), (100, 300)])
    table = G.t_scheduler_with_dual_gaussian_pdf(self, W, 2400, 799)
    np.savez(os.path.join(OUT, "ahds_schedule.npz"), pdf=np.asarray(W, dtype=np.float64), table=np.asarray(table, dtype=np.int64),
             args="init=[260,60,280] ratios=[0.41,0.21,0.375] ranges=[(0,350),(350,450),(450,800)] total=800 "
                  "bounds=[(200,400),(20,100),(100,300)] N=2400 t0=799")
    t = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    c = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    np.savez(os.path.join(OUT, "rescale_noise_cfg.npz"), noise_cfg=c.numpy(), noise_pred_text=t.numpy(),
             out=mod.rescale_noise_cfg(c, t, guidance_rescale=0.7).numpy(), guidance_rescale=0.7)
    print("AHDS table written:", table[:5], table[-5:])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ahds":
        ahds()
    else:
        main()
