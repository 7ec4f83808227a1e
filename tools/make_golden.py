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


class _Stub(types.ModuleType):
    """Permissive stand-in for a package that is absent here: attribute access fabricates
    child stubs, calling it is an identity decorator, subscripting returns itself, usable as a base class."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        child = _Stub(self.__name__ + "." + name)
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


def _install_threestudio_stubs():
    import typing
    names = ["threestudio", "threestudio.utils", "threestudio.utils.base", "threestudio.utils.misc", "threestudio.utils.typing",
             "threestudio.utils.ops", "threestudio.models", "threestudio.models.prompt_processors",
             "threestudio.models.prompt_processors.base",
             "threestudio.models.guidance", "threestudio.models.guidance.models", "threestudio.models.guidance.models.ip_adapter",
             "threestudio.models.guidance.models.ip_adapter.ip_adapter_faceid", "threestudio.models.guidance.models.pipeline_ipa",
             "threestudio.models.guidance.models.pipeline_ipa_controlnet", "diffusers", "diffusers.utils", "diffusers.utils.import_utils",
             "diffusers.models", "diffusers.models.lora", "pytorch_lightning", "pytorch_lightning.utilities",
             "pytorch_lightning.utilities.rank_zero",
             "cv2", "insightface", "insightface.app", "insightface.utils", "PIL", "PIL.Image", "tqdm"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _Stub(n)
    ty = sys.modules["threestudio.utils.typing"]
    for k in dir(typing):
        if not k.startswith("_"):
            setattr(ty, k, getattr(typing, k))
    ty.Tensor = torch.Tensor
    jt = ["Bool", "Complex", "Float", "Inexact", "Int", "Integer", "Num", "Shaped", "UInt", "DictConfig", "typechecker"]
    for k in jt:
        setattr(ty, k, _Stub("jaxtyping_stub"))
    ty.__all__ = [k for k in dir(typing) if not k.startswith("_")] + ["Tensor"] + jt

    class BaseObject:
        class Config:
            pass
    sys.modules["threestudio.utils.base"].BaseObject = BaseObject


def _load_by_path(module_name, rel_path, package):
    path = os.path.join(REF, rel_path)
    spec = importlib.util.spec_from_file_location(module_name, path)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = package
    spec.loader.exec_module(mod)
    return mod


def _load_reference_guidance():
    _install_threestudio_stubs()
    return _load_by_path("threestudio.models.guidance.ipa_guidance", "threestudio/models/guidance/ipa_guidance.py",
                         "threestudio.models.guidance")


def ahds():
    """(5) AHDS timestep table: loads threestudio/models/guidance/ipa_guidance.py by path with permissive stubs
     and calls the two schedule functions unbound."""
    mod = _load_reference_guidance()
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


def guidance():
    """Reference-derived fixtures for the guidance math (outputs only; the inputs come from tests/fixture_inputs.py):
      attention_processors.npz  LoRAAttnProcessor2_0 'normal' + 'refine' states and LoRAIPAttnProcessor2_0
                                (ip_adapter/attention_processor_faceid.py:211-395, 398-523)
      sds_grad.npz              compute_grad_anpg / compute_grad_sds (ipa_guidance.py:361-519) around a stand-in forward_unet
      prompt_directions.npz     the 13 view-dependent prompt directions + PromptProcessorOutput.get_text_embeddings
                                (prompt_processors/base.py:52-81, 252-335)
      refine_timesteps.npz      the timestep expression of refine.py:176-178
    Absent third-party pieces restated as stand-ins, by their published behaviour (diffusers 0.27, not vendored):
    diffusers.models.lora.LoRALinearLayer (down / up bias-free linears, optional network_alpha / rank factor) and
    DDIMScheduler.add_noise (sqrt(acp_t) x + sqrt(1 - acp_t) eps on the scaled-linear 0.00085 -> 0.012 betas)."""
    import torch.nn as nn
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import fixture_inputs as fx
    _install_threestudio_stubs()

    class LoRALinearLayer(nn.Module):
        def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
            super().__init__()
            self.down = nn.Linear(in_features, rank, bias=False)
            self.up = nn.Linear(rank, out_features, bias=False)
            self.network_alpha, self.rank = network_alpha, rank

        def forward(self, hidden_states):
            orig = hidden_states.dtype
            up = self.up(self.down(hidden_states.to(self.down.weight.dtype)))
            if self.network_alpha is not None:
                up = up * (self.network_alpha / self.rank)
            return up.to(orig)
    sys.modules["diffusers.models.lora"].LoRALinearLayer = LoRALinearLayer
    ap = _load_by_path("threestudio.models.guidance.models.ip_adapter.attention_processor_faceid",
                       "threestudio/models/guidance/models/ip_adapter/attention_processor_faceid.py",
                       "threestudio.models.guidance.models.ip_adapter")

    T = torch.from_numpy

    def fake_attn(w, dim, ctx_dim, heads):
        """The fields of diffusers' Attention the processors read (SD1.5 transformer blocks: no norms, no residual)."""
        lin = lambda W, b=None: (lambda x: torch.nn.functional.linear(x, T(W), None if b is None else T(b)))  # noqa: E731
        return SimpleNamespace(spatial_norm=None, group_norm=None, norm_cross=False, residual_connection=False,
                               rescale_output_factor=1.0, heads=heads, prepare_attention_mask=lambda m, n, b: None,
                               to_q=lin(w["to_q"]), to_k=lin(w["to_k"]), to_v=lin(w["to_v"]),
                               to_out=[lin(w["to_out_w"], w["to_out_b"]), lambda x: x])

    def load_lora(proc, w):
        with torch.no_grad():
            for n in ("q", "k", "v", "out"):
                layer = getattr(proc, "to_%s_lora" % n)
                layer.down.weight.copy_(T(w["lora_%s_down" % n]))
                layer.up.weight.copy_(T(w["lora_%s_up" % n]))

    out = {}
    dim, heads, rank, B = 320, 8, 128, 2
    with torch.no_grad():
        # --- self-attention, 'normal' state (stage 1), 64 and 256 tokens
        w = fx.attn_weights(11, dim, None, rank)
        proc = ap.LoRAAttnProcessor2_0(name="p", target_processor_names=[], state="normal", stored_zt={}, hidden_size=dim,
                                       cross_attention_dim=None, rank=rank)
        load_lora(proc, w)
        attn = fake_attn(w, dim, None, heads)
        for n_tok, nb in ((64, B), (256, 1)):
            out["self_normal_%d" % n_tok] = proc(attn, T(fx.attn_tokens(21 + n_tok, nb, n_tok, dim))).numpy()
        # --- self-attention, 'refine' state: canonical views store tokens, k-views attend mutually, v-views blend
        n_tok, nb = fx.REFINE_TOKENS, 1
        names = ["tgt"]
        proc = ap.LoRAAttnProcessor2_0(name="tgt", target_processor_names=names, state="refine", stored_zt={},
                                       total_denoise_step=fx.REFINE_STEPS, lambda_self=fx.REFINE_LAMBDA_SELF, hidden_size=dim,
                                       cross_attention_dim=None, rank=rank)
        load_lora(proc, w)
        other = ap.LoRAAttnProcessor2_0(name="other", target_processor_names=names, state="refine", stored_zt={},
                                        total_denoise_step=fx.REFINE_STEPS, hidden_size=dim, cross_attention_dim=None, rank=rank)
        load_lora(other, w)
        for vi, (view, pair, weights) in enumerate(fx.REFINE_VIEWS):
            for p_ in (proc, other):
                p_.cur_view_name = view
                p_.stored_zt[view] = []                                   # refine.py:205-207
                if "v" in view:
                    p_.cur_key_view_name_pair, p_.cur_key_view_weight_pair = pair, weights
            for step in range(fx.REFINE_STEPS):
                x = T(fx.refine_tokens(vi, step, nb, n_tok, dim))
                out["refine_%s_%d" % (view, step)] = proc(attn, x).numpy()
                if view == "front":      # a non-target layer in the refine state only works for the key views in the reference
                    out["refine_other_%s_%d" % (view, step)] = other(attn, x).numpy()
        # --- decoupled text / image cross-attention
        wc = fx.attn_weights(12, dim, 768, rank, ip=True)
        procx = ap.LoRAIPAttnProcessor2_0(hidden_size=dim, cross_attention_dim=768, rank=rank, scale=0.5, num_tokens=4)
        load_lora(procx, wc)
        procx.to_k_ip.weight.copy_(T(wc["to_k_ip"]))
        procx.to_v_ip.weight.copy_(T(wc["to_v_ip"]))
        attnx = fake_attn(wc, dim, 768, heads)
        for n_tok, nb in ((64, B), (256, 1)):
            out["cross_ip_%d" % n_tok] = procx(attnx, T(fx.attn_tokens(31 + n_tok, nb, n_tok, dim)),
                                               encoder_hidden_states=T(fx.attn_tokens(41, nb, 81, 768))).numpy()
    np.savez_compressed(os.path.join(OUT, "attention_processors.npz"), dim=dim, heads=heads, rank=rank, batch=B, ip_scale=0.5, **out)

    # ---------------- view-dependent prompt directions ----------------
    rz = sys.modules["pytorch_lightning.utilities.rank_zero"]
    rz.rank_zero_only = lambda f: f
    real_tf = sys.modules.get("transformers")
    sys.modules["transformers"] = _Stub("transformers")      # base.py:9 imports BertForMaskedLM (prompt debiasing, unused)
    try:
        base = _load_by_path("threestudio.models.prompt_processors.base", "threestudio/models/prompt_processors/base.py",
                             "threestudio.models.prompt_processors")
    finally:
        if real_tf is not None:
            sys.modules["transformers"] = real_tf
        else:
            del sys.modules["transformers"]

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    pp = object.__new__(base.PromptProcessor)
    prompt = "a person wearing a coat"
    pp.cfg = Cfg(prompt=prompt, negative_prompt="ugly", negative_prompt_faceid="blurry", null_prompt="", head_offset=0.65,
                 view_dependent_prompt_front=False, use_prompt_debiasing=False, use_cache=False, spawn=False,
                 use_ipa_faceid=True, pretrained_realistic_model_name_or_path="m", pretrained_sd_model_name_or_path="m")
    cwd = os.getcwd()
    os.chdir(REF)                              # configure() opens load/prompt_library.json relative to the working directory
    try:
        try:
            pp.configure()
        except NotImplementedError:            # spawn_func of the base class: the text encoder is out of reach; the
            pass                               # directions / prompts are already built at that point
    finally:
        os.chdir(cwd)
    case = fx.sds_case(7)
    el, az, cent, vis, dist = [T(a) for a in fx.direction_grid()]
    marker = torch.arange(13, dtype=torch.float32).view(13, 1, 1).expand(13, 2, 3).contiguous()      # row i holds the value i
    ppo = base.PromptProcessorOutput(
        text_embeddings=torch.full((1, 2, 3), 100.0), uncond_text_embeddings=torch.full((1, 2, 3), 200.0),
        null_embeddings=torch.full((1, 2, 3), 300.0), text_embeddings_vd=marker, uncond_text_embeddings_vd=marker + 1000.0,
        directions=pp.directions, direction2idx=pp.direction2idx, use_perp_neg=False, perp_neg_f_sb=(1, 0.5, -0.606),
        perp_neg_f_fsb=(1, 0.5, 0.967), perp_neg_f_fs=(4, 0.5, -2.426), perp_neg_f_sf=(4, 0.5, -2.426))
    emb = ppo.get_text_embeddings(el, az, cent, vis, dist, True)
    n = el.shape[0]
    emb_flat = ppo.get_text_embeddings(el, az, cent, vis, dist, False)
    np.savez(os.path.join(OUT, "prompt_directions.npz"), names=np.array([d.name for d in pp.directions]),
             direction2idx_keys=np.array(list(pp.direction2idx.keys())), direction2idx_values=np.array(list(pp.direction2idx.values())),
             prompt=prompt, prompts_vd=np.array(pp.prompts_vd), negative_prompts_vd=np.array(pp.negative_prompts_vd),
             direction_idx=emb[:n, 0, 0].numpy().astype(np.int64), uncond_idx=(emb[n:2 * n, 0, 0] - 1000.0).numpy().astype(np.int64),
             null_value=emb[2 * n:, 0, 0].numpy(), not_view_dependent=emb_flat[:, 0, 0].numpy())

    # ---------------- compute_grad_anpg / compute_grad_sds ----------------
    mod = _load_reference_guidance()
    G = mod.StableDiffusionGuidance
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)

    class Scheduler:                                  # stand-in for diffusers.DDIMScheduler (configured at ipa_guidance.py:139-147)
        def add_noise(self, x, noise, t):
            a = acp.to(x.dtype)[t]
            return a.sqrt().view(-1, 1, 1, 1) * x + (1 - a).sqrt().view(-1, 1, 1, 1) * noise
    ppo_real = base.PromptProcessorOutput(
        text_embeddings=T(case["text"]), uncond_text_embeddings=T(case["uncond"]), null_embeddings=T(case["null"]),
        text_embeddings_vd=T(case["text_vd"]), uncond_text_embeddings_vd=T(case["uncond_vd"]), directions=pp.directions,
        direction2idx=pp.direction2idx, use_perp_neg=False, perp_neg_f_sb=(1, 0.5, -0.606), perp_neg_f_fsb=(1, 0.5, 0.967),
        perp_neg_f_fs=(4, 0.5, -2.426), perp_neg_f_sf=(4, 0.5, -2.426))
    res = {}
    for tag, vd, clip, weighting, rescale in (("anpg", True, True, "sds", 0.0), ("anpg_flat_noclip", False, False, "fantasia3d", 0.0),
                                              ("sds", True, True, "sds", 0.0), ("sds_rescale", True, False, "uniform", 0.75)):
        me = object.__new__(G)
        me.cfg = SimpleNamespace(view_dependent_prompting=vd, guidance_scale=7.5, weighting_strategy=weighting,
                                 grad_clip_pixel=clip, grad_clip_threshold=1.0, guidance_rescale=rescale)
        me.alphas, me.scheduler = acp, Scheduler()
        me.pos_image_embeds, me.neg_image_embeds, me.null_image_embeds = T(case["pos_image"]), T(case["neg_image"]), T(case["null_image"])
        neg = torch.cat([T(case["uncond"]).expand(4, -1, -1), me.neg_image_embeds], dim=1)
        pos = torch.cat([T(case["text"]).expand(4, -1, -1), me.pos_image_embeds], dim=1)
        null = torch.cat([T(case["null"]).expand(4, -1, -1), me.null_image_embeds], dim=1)
        me.final_prompt_embeds_npn = torch.cat([neg, pos, null])       # as prepare_for_sds builds them (:296-308)
        me.final_prompt_embeds_np = torch.cat([neg, pos])
        me.forward_unet = fx.fake_forward_unet
        fn = G.compute_grad_anpg if tag.startswith("anpg") else G.compute_grad_sds
        torch.manual_seed(2024)
        grad, util = fn(me, T(case["latents"]), T(case["control"]), T(case["t"]), ppo_real, True, T(case["all_vis_all"]),
                        T(case["elevation"]), T(case["azimuth"]), T(case["center"]), T(case["camera_distances"]))
        res[tag + "_grad"] = grad.numpy()
        res[tag + "_latents_noisy"] = util["latents_noisy"].numpy()
    np.savez_compressed(os.path.join(OUT, "sds_grad.npz"), seed=2024, case_seed=7, **res)

    # ---------------- refine timesteps (refine.py:176-178) ----------------
    ts = torch.linspace(0, 999, 50, dtype=torch.int64).round().flip(dims=[0])
    np.savez(os.path.join(OUT, "refine_timesteps.npz"), timesteps=ts.numpy(), timesteps_sub=ts[-8:].numpy())
    print("guidance fixtures written; refine timesteps:", ts[-8:].tolist())
    print("direction2idx:", pp.direction2idx)


def backward():
    """Reference-AUTOGRAD fixtures for the two per-Gaussian backward stages of the rasterizer that the reference also
    states in Python: SH colour (gaussian_renderer/__init__.py:71-78 + sh_utils.eval_sh) and the 3-D covariance
    (gaussian_model.py:16-20 + general_utils.build_scaling_rotation / strip_symmetric).  Widen the oracle's pinned
    perimeter into its backward (oracle_sh_backward / oracle_cov3D_backward)."""
    _install_device_shim()
    _install_stub_modules()
    from gaussiansplatting.utils import sh_utils, general_utils
    rng = np.random.default_rng(77)
    Pn = 128
    xyz = rng.uniform(-0.5, 0.5, (Pn, 3)).astype(np.float32)
    campos = np.array([0.3, -1.4, 0.5], np.float32)
    feats = (rng.standard_normal((Pn, 16, 3)) * 0.4).astype(np.float32)       # model layout [P, K, 3] (gaussian_model.py:97-100)
    feats[:, 0] *= 4.0                                                        # some colours below zero: the clamp has both branches
    gcol = rng.standard_normal((Pn, 3)).astype(np.float32)
    out = dict(xyz=xyz, campos=campos, features=feats, gcol=gcol)
    for deg in range(4):
        K = (deg + 1) ** 2
        f = torch.from_numpy(feats[:, :K].copy()).requires_grad_(True)
        x = torch.from_numpy(xyz).requires_grad_(True)
        shs_view = f.transpose(1, 2).view(-1, 3, K)
        dir_pp = x - torch.from_numpy(campos).repeat(Pn, 1)
        dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(sh_utils.eval_sh(deg, shs_view, dir_n) + 0.5, 0.0)
        (rgb * torch.from_numpy(gcol)).sum().backward()
        out["rgb%d" % deg] = rgb.detach().numpy()
        out["dfeatures%d" % deg] = f.grad.numpy()
        out["dxyz%d" % deg] = np.zeros_like(xyz) if x.grad is None else x.grad.numpy()    # degree 0 ignores the direction
    s_ = (rng.uniform(0.002, 0.08, (Pn, 3))).astype(np.float32)
    q_ = rng.standard_normal((Pn, 4))
    q_ = (q_ / np.linalg.norm(q_, axis=1, keepdims=True)).astype(np.float32)
    dcov = rng.standard_normal((Pn, 6)).astype(np.float32)
    out.update(scales=s_, rotations=q_, dcov=dcov)
    for tag, mod in (("m1", 1.0), ("m17", 1.7)):
        s = torch.from_numpy(s_).requires_grad_(True)
        q = torch.from_numpy(q_).requires_grad_(True)
        L = general_utils.build_scaling_rotation(mod * s, q)
        cov6 = general_utils.strip_symmetric(L @ L.transpose(1, 2))
        (cov6 * torch.from_numpy(dcov)).sum().backward()
        out["cov6_" + tag] = cov6.detach().numpy()
        out["dscales_" + tag] = s.grad.numpy()
        out["drotations_" + tag] = q.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "preprocess_backward.npz"), **out)
    print("backward fixtures written")


def ops():
    """threestudio/utils/ops.py:266-300 (get_projection_matrix, get_mvp_matrix, binary_cross_entropy) — the data module's
    mvp matrices (camera_data.py:462-463) that the pose-map drawing consumes (GaussianIP.py:177)."""
    _install_threestudio_stubs()
    sys.modules.setdefault("igl", _Stub("igl"))
    mod = _load_by_path("threestudio.utils.ops_by_path", "threestudio/utils/ops.py", "threestudio.utils")
    rng = np.random.default_rng(5)
    c2w = torch.stack([orbit_c2w(rng.uniform(-30, 30), rng.uniform(-180, 180), rng.uniform(1.3, 1.7)) for _ in range(8)])
    fovy = torch.tensor(np.radians(rng.uniform(40, 70, 8)), dtype=torch.float32)
    proj = mod.get_projection_matrix(fovy, 1.0, 0.1, 1000.0)
    proj_wide = mod.get_projection_matrix(fovy, 1.5, 0.1, 1000.0)
    mvp = mod.get_mvp_matrix(c2w, proj)
    x = torch.rand(64, generator=torch.Generator().manual_seed(3)) * 0.98 + 0.01
    np.savez(os.path.join(OUT, "mvp.npz"), c2w=c2w.numpy(), fovy=fovy.numpy(), proj=proj.numpy(), proj_aspect_1p5=proj_wide.numpy(),
             mvp=mvp.numpy(), bce_x=x.numpy(), bce=mod.binary_cross_entropy(x, x).numpy())
    print("ops fixtures written")


def ply():
    """The vertex table GaussianModel.save_ply hands to plyfile (gaussian_model.py:190-216): attribute names in order
    and the float32 rows, captured at the PlyElement.describe boundary (plyfile itself is an absent third-party package:
    it then writes "ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... / end_header"
    followed by the rows — its published format)."""
    _install_device_shim()
    _install_stub_modules()
    captured = {}

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            captured["elements"], captured["name"] = elements, name
            return elements

    class PlyData:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = path
    sys.modules["plyfile"].PlyElement, sys.modules["plyfile"].PlyData = PlyElement, PlyData
    import gaussiansplatting.scene.gaussian_model as gmod
    gmod.PlyElement, gmod.PlyData = PlyElement, PlyData
    gmod.mkdir_p = lambda p: None
    from gaussiansplatting.utils.graphics_utils import BasicPointCloud
    out = {}
    for deg in (0, 2):
        rng = np.random.default_rng(9 + deg)
        gm = gmod.GaussianModel(deg)
        n = 40
        gm.create_from_pcd(BasicPointCloud(rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32), rng.uniform(0, 1, (n, 3)).astype(np.float32), None), 4.0)
        with torch.no_grad():
            gm._features_rest += torch.from_numpy(rng.standard_normal(tuple(gm._features_rest.shape)).astype(np.float32))
            gm._rotation += torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)) * 0.2
        gm.save_ply("/tmp/unused.ply")
        el = captured["elements"]
        out["names_deg%d" % deg] = np.array(el.dtype.names)
        out["table_deg%d" % deg] = np.stack([el[k] for k in el.dtype.names], axis=1).astype(np.float32)
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            out["deg%d%s" % (deg, k)] = getattr(gm, k).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ply_layout.npz"), element_name=captured["name"], **out)
    print("ply layout written:", list(out["names_deg0"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ahds":
        ahds()
    elif len(sys.argv) > 1 and sys.argv[1] == "guidance":
        guidance()
    elif len(sys.argv) > 1 and sys.argv[1] == "backward":
        backward()
    elif len(sys.argv) > 1 and sys.argv[1] == "ops":
        ops()
    elif len(sys.argv) > 1 and sys.argv[1] == "ply":
        ply()
    else:
        main()
