"""Gaussian parameter store + Adam-state surgery + densify / prune, API-compatible with the reference.

Reference: gaussiansplatting/scene/gaussian_model.py — activations :15-30, fields :33-48, getters :84-107,
create_from_pcd :113-136, training_setup :138-159, lr schedule :161-181, PLY I/O :183-264, optimizer surgery
:266-355, densify_and_split :357-380, densify_and_clone :382-393, densify_and_prune :395-410, prune_only :413-418,
add_densification_stats :420-422.  Pinned by tests/golden/gaussian_model.npz (traces captured from the imported
reference class).

Differences that do not change results: tensors follow `device` (default "cuda") instead of a hard-coded "cuda";
the six parameter groups are handled through one table instead of six hand-written copies; PLY files are read and
written by a small built-in binary reader/writer (same header and column order as plyfile produces).
"""
import os

import numpy as np
import torch
from torch import nn

from ..utils.general import (build_rotation, build_scaling_rotation, get_expon_lr_func, inverse_sigmoid,
                             strip_symmetric)
from ..utils.graphics import BasicPointCloud
from ..utils.sh import RGB2SH

# optimizer group name -> attribute holding the parameter (order = the reference's param_groups order)
_GROUPS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
           ("scaling", "_scaling"), ("rotation", "_rotation"))


def _covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    L = build_scaling_rotation(scaling_modifier * scaling, rotation)
    return strip_symmetric(L @ L.transpose(1, 2))


_FUSED_ACT = True     # False: the three getters' op chains (the same-box A/B of DESIGN §4d; tests flip the attribute)


class _Activate(torch.autograd.Function):
    """(sigmoid(opacity), exp(scaling), normalize(rotation)) in one launch, their backward in one (include/gip_model.h:
    gip_activate_gaussians*): the values of the getters :72-89, which stay what every other caller uses."""

    @staticmethod
    def forward(ctx, o, s, q):
        import ctypes

        from .. import _lib
        oo, so, qo = torch.empty_like(o), torch.empty_like(s), torch.empty_like(q)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        rc = _lib.model_lib().gip_activate_gaussians(p(o), p(s), p(q), o.shape[0], p(oo), p(so), p(qo),
                                                     ctypes.c_void_p(torch.cuda.current_stream(o.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_activate_gaussians failed with status %d" % rc)
        ctx.save_for_backward(oo, so, q)
        return oo, so, qo

    @staticmethod
    def backward(ctx, go, gs, gq):
        import ctypes

        from .. import _lib
        oo, so, q = ctx.saved_tensors
        p = lambda t: ctypes.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
        go, gs, gq = (None if g_ is None else g_.contiguous() for g_ in (go, gs, gq))
        need = ctx.needs_input_grad
        d_o = torch.empty_like(oo) if need[0] else None
        d_s = torch.empty_like(so) if need[1] else None
        d_q = torch.empty_like(q) if need[2] else None
        rc = _lib.model_lib().gip_activate_gaussians_backward(p(oo), p(so), p(q), p(go), p(gs), p(gq), oo.shape[0], p(d_o), p(d_s), p(d_q),
                                                              ctypes.c_void_p(torch.cuda.current_stream(oo.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_activate_gaussians_backward failed with status %d" % rc)
        return d_o, d_s, d_q


class GaussianModel:
    def get_activated(self):
        """(get_opacity, get_scaling, get_rotation) of one step — on the GPU with the stock activations one launch forward and
        one backward instead of the three getters' op chains."""
        o, s, q = self._opacity, self._scaling, self._rotation
        if (_FUSED_ACT and o.is_cuda and self.opacity_activation is torch.sigmoid and self.scaling_activation is torch.exp and
                self.rotation_activation is torch.nn.functional.normalize and
                all(t.dtype == torch.float32 and t.is_contiguous() for t in (o, s, q)) and o.shape[0] > 0):
            return _Activate.apply(o, s, q)
        return self.get_opacity, self.get_scaling, self.get_rotation

    def setup_functions(self):
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.covariance_activation = _covariance_from_scaling_rotation
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree: int, device="cuda"):
        self.device = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        for _, attr in _GROUPS:
            setattr(self, attr, torch.empty(0))
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.setup_functions()

    # ------------------------------------------------------------------ state hand-off
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
         self._opacity, self.max_radii2D, grad_accum, denom, opt_dict, self.spatial_lr_scale) = model_args
        self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = grad_accum, denom
        self.optimizer.load_state_dict(opt_dict)

    # ------------------------------------------------------------------ activated views
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        # raw (un-normalised) rotation: build_rotation normalises it (reference quirk kept, :106-107)
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ initialisation
    def create_from_pcd(self, pcd: BasicPointCloud, spatial_lr_scale: float, dist2=None):
        """`dist2` (mean squared 3-NN distance per point) defaults to the HIP distCUDA2 replacement."""
        dev = self.device
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        n = pts.shape[0]
        n_coef = (self.max_sh_degree + 1) ** 2
        feats = torch.zeros((n, 3, n_coef), dtype=torch.float32, device=dev)
        feats[:, :3, 0] = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(dev))
        print("Number of points at initialisation : ", n)
        if dist2 is None:
            from ..knn import distCUDA2
            dist2 = distCUDA2(pts)
        dist2 = torch.clamp_min(torch.as_tensor(dist2, dtype=torch.float32, device=dev), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float32, device=dev))
        self._xyz = nn.Parameter(pts.contiguous().requires_grad_(True))      # (a point array that arrives transposed keeps its strides otherwise)
        self._features_dc = nn.Parameter(feats[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(feats[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opac.requires_grad_(True))
        self.max_radii2D = torch.zeros((n,), device=dev)

    def training_setup(self, training_args, fused: bool = False):
        """`fused=True` (GPU only): the same update rule as torch.optim.Adam with all six groups in ONE launch
        (scene/adam.py -> gip_adam_step; GIP_ADAM=torch selects torch's fused Adam, 18 launches), and the form that lets a
        GradScaler skip on the device: `scaler.step()` then issues no host synchronisation (the reference's plain Adam
        under `precision: 16-mixed` pays one `.item()` per step)."""
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.percent_dense = training_args.percent_dense
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        lrs = {"xyz": training_args.position_lr_init * self.spatial_lr_scale, "f_dc": training_args.feature_lr,
               "f_rest": training_args.feature_lr / 20.0, "opacity": training_args.opacity_lr,
               "scaling": training_args.scaling_lr, "rotation": training_args.rotation_lr}
        self.params_list = [{"params": [getattr(self, attr)], "lr": lrs[name], "name": name} for name, attr in _GROUPS]
        if fused and dev.type == "cuda":
            from .adam import GipAdam          # the same update as torch's fused Adam, all six groups in ONE launch
            self.optimizer = GipAdam(self.params_list, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(self.params_list, lr=0.0, eps=1e-15, **({"fused": True} if fused else {}))
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)

    def set_refine_learning_rate(self, iteration):
        self.update_learning_rate(iteration)

    # ------------------------------------------------------------------ PLY (float32 columns, plyfile-compatible)
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += ["f_dc_%d" % i for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += ["f_rest_%d" % i for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += ["scale_%d" % i for i in range(self._scaling.shape[1])]
        names += ["rot_%d" % i for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        cols = [xyz, np.zeros_like(xyz),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._opacity.detach().cpu().numpy(), self._scaling.detach().cpu().numpy(),
                self._rotation.detach().cpu().numpy()]
        table = np.concatenate(cols, axis=1).astype("<f4")
        names = self.construct_list_of_attributes()
        assert table.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
        header += "".join("property float %s\n" % n for n in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(np.ascontiguousarray(table).tobytes())

    @staticmethod
    def _read_ply(path):
        with open(path, "rb") as f:
            names, n, fmt = [], 0, None
            while True:
                line = f.readline().decode("ascii").strip()
                if line.startswith("format"):
                    fmt = line.split()[1]
                elif line.startswith("element vertex"):
                    n = int(line.split()[-1])
                elif line.startswith("property"):
                    parts = line.split()
                    if parts[1] not in ("float", "float32"):
                        raise ValueError("only float32 vertex properties are supported: %s" % line)
                    names.append(parts[2])
                elif line == "end_header":
                    break
            if fmt != "binary_little_endian":
                raise ValueError("unsupported PLY format %r" % fmt)
            data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
        return {name: data[:, i] for i, name in enumerate(names)}

    def load_ply(self, path):
        col = self._read_ply(path)
        dev = self.device
        xyz = np.stack((col["x"], col["y"], col["z"]), axis=1)
        n = xyz.shape[0]

        def numbered(prefix):
            keys = sorted((k for k in col if k.startswith(prefix)), key=lambda k: int(k.split("_")[-1]))
            return np.stack([col[k] for k in keys], axis=1) if keys else np.zeros((n, 0), np.float32)

        f_dc = np.stack((col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]), axis=1).reshape(n, 3, 1)
        f_rest = numbered("f_rest_")
        assert f_rest.shape[1] == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_rest = f_rest.reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1)

        def param(a):
            return nn.Parameter(torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=dev).requires_grad_(True))

        self._xyz = param(xyz)
        self._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(f_rest, dtype=torch.float, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = param(col["opacity"][..., None])
        self._scaling = param(numbered("scale_"))
        self._rotation = param(numbered("rot"))
        self.active_sh_degree = self.max_sh_degree

    # ------------------------------------------------------------------ optimizer surgery
    def _rebuild(self, transform, state_transform):
        """Replace every group's parameter by transform(name, old) and its Adam moments by state_transform(name, m);
        returns {name: new parameter}."""
        out = {}
        for group in self.optimizer.param_groups:
            assert len(group["params"]) == 1
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(transform(group["name"], old).requires_grad_(True))
            if state is not None:
                state["exp_avg"] = state_transform(group["name"], state["exp_avg"])
                state["exp_avg_sq"] = state_transform(group["name"], state["exp_avg_sq"])
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
        return out

    def _adopt(self, tensors):
        for name, attr in _GROUPS:
            if name in tensors:
                setattr(self, attr, tensors[name])

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] != name:
                continue
            state = self.optimizer.state.pop(group["params"][0], None)
            state["exp_avg"] = torch.zeros_like(tensor)
            state["exp_avg_sq"] = torch.zeros_like(tensor)
            group["params"][0] = nn.Parameter(tensor.requires_grad_(True))
            self.optimizer.state[group["params"][0]] = state
            out[name] = group["params"][0]
        return out

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    def _prune_optimizer(self, mask):
        return self._rebuild(lambda _n, p: p[mask], lambda _n, m: m[mask])

    def prune_points(self, mask):
        keep = ~mask
        self._adopt(self._prune_optimizer(keep))
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def cat_tensors_to_optimizer(self, tensors_dict):
        return self._rebuild(lambda n, p: torch.cat((p, tensors_dict[n]), dim=0),
                             lambda n, m: torch.cat((m, torch.zeros_like(tensors_dict[n])), dim=0))

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation):
        self._adopt(self.cat_tensors_to_optimizer({"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest,
                                                   "opacity": new_opacities, "scaling": new_scaling,
                                                   "rotation": new_rotation}))
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)

    # ------------------------------------------------------------------ densification
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        padded = torch.zeros((n,), device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._features_dc[sel].repeat(N, 1, 1), self._features_rest[sel].repeat(N, 1, 1),
                                   self._opacity[sel].repeat(N, 1), new_scaling, self._rotation[sel].repeat(N, 1))
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool))))

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & \
              (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel], self._opacity[sel],
                                   self._scaling[sel], self._rotation[sel])

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, max_world_size):
        if self._xyz.is_cuda:
            return self._densify_and_prune_fused(max_grad, min_opacity, extent, max_screen_size, max_world_size)
        return self._densify_and_prune_stepwise(max_grad, min_opacity, extent, max_screen_size, max_world_size)

    def _gather_all(self, index, n_old, new_rows):
        """One launch (include/gip_model.h) rebuilds the six parameters and their Adam moments:
        row j <- old[index[j]] if index[j] < n_old else new_rows[name][index[j] - n_old] (moments of new rows: zeros)."""
        import ctypes
        from .. import _lib
        lib = _lib.model_lib()
        n_out = int(index.numel())
        descs, outs, keep_alive = [], {}, []
        for group in self.optimizer.param_groups:
            name, old = group["name"], group["params"][0]
            new = new_rows[name].contiguous()
            dst = torch.empty((n_out,) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
            rb = dst[0:1].numel() * dst.element_size() if n_out else 0     # zero for SH degree 0's empty f_rest rows
            if rb:
                descs.append((old.data.contiguous(), new, dst, rb))
            state = self.optimizer.state.get(old, None)
            moments = {}
            if state is not None and "exp_avg" in state:
                for key in ("exp_avg", "exp_avg_sq"):
                    mdst = torch.empty_like(dst)
                    if rb:
                        descs.append((state[key].contiguous(), None, mdst, rb))
                    moments[key] = mdst
            outs[name] = (dst, moments)
        arr = (_lib.GipGatherTensor * len(descs))()
        for i, (o, nw, d, rb) in enumerate(descs):
            arr[i].old_rows = o.data_ptr() if o.numel() else None
            arr[i].new_rows = nw.data_ptr() if (nw is not None and nw.numel()) else None
            arr[i].dst, arr[i].row_bytes = d.data_ptr(), rb
            keep_alive.append((o, nw, d))
        rc = 0
        if descs:
            rc = lib.gip_gather_rows(arr, len(descs), ctypes.c_void_p(index.data_ptr()), n_out, n_old,
                                     ctypes.c_void_p(torch.cuda.current_stream(index.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_gather_rows failed with status %d" % rc)
        rebuilt = {}
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            dst, moments = outs[group["name"]]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(dst.requires_grad_(True))
            if state is not None:
                state.update(moments)
                self.optimizer.state[new] = state
            group["params"][0] = new
            rebuilt[group["name"]] = new
        return rebuilt

    # ------------------------------------------------------------------ spatial order (not in the reference)
    @staticmethod
    def morton_order(xyz, bits=10):
        """Permutation that sorts points along a 3-D Morton (Z-order) curve.  Rendering is invariant under a permutation
        of the Gaussians (up to float summation order); a spatially coherent order lets the 256 Gaussians of a
        preprocess workgroup share tile-histogram atomics and keeps a tile's records close in memory."""
        lo, hi = xyz.min(dim=0).values, xyz.max(dim=0).values
        q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * (2 ** bits - 1)).long().clamp_(0, 2 ** bits - 1)
        code = torch.zeros(xyz.shape[0], dtype=torch.long, device=xyz.device)
        for b in range(bits):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return torch.argsort(code)

    def sort_spatially(self):
        """Re-order the Gaussians (parameters, Adam moments, densification statistics) along a Morton curve.  Call after
        create_from_pcd / densify_and_prune.  render() is unchanged up to float summation order and the order of entries
        with identical depth bits (ties are broken by Gaussian index, as in the reference)."""
        perm = self.morton_order(self._xyz.detach())
        if self.optimizer is not None and self._xyz.is_cuda:
            P = self._xyz.shape[0]
            empty = {g["name"]: g["params"][0].detach()[:0] for g in self.optimizer.param_groups}
            self._adopt(self._gather_all(perm.contiguous(), P, empty))
        else:
            for _, attr in _GROUPS:
                setattr(self, attr, nn.Parameter(getattr(self, attr).detach()[perm].requires_grad_(True)))
            if self.optimizer is not None:
                raise ValueError("sort_spatially on CPU must run before training_setup")
        for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(self, name)
            if t.numel() and t.shape[0] == perm.shape[0]:
                setattr(self, name, t[perm])
        return perm

    def _densify_and_prune_fused(self, max_grad, min_opacity, extent, max_screen_size, max_world_size, N=2):
        """densify_and_clone -> densify_and_split -> prune_points x2 of the reference (gaussian_model.py:357-411) with
        the same selections, the same torch.normal draw and the same final row order, but every tensor rebuilt ONCE:
        the survivors of [old | clones | split children] are described by one index list and moved by gip_gather_rows."""
        P, dev = self.get_xyz.shape[0], self.get_xyz.device
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        scaling = self.get_scaling
        big = scaling.max(dim=1).values > self.percent_dense * extent
        clone_sel = (torch.norm(grads, dim=-1) >= max_grad) & ~big
        split_sel = (grads.squeeze(-1) >= max_grad) & big
        stds = scaling[split_sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = build_rotation(self._rotation[split_sel]).repeat(N, 1, 1)
        child_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[split_sel].repeat(N, 1)
        child_scaling = self.scaling_inverse_activation(scaling[split_sel].repeat(N, 1) / (0.8 * N))
        new_rows = {
            "xyz": torch.cat((self._xyz[clone_sel], child_xyz)),
            "f_dc": torch.cat((self._features_dc[clone_sel], self._features_dc[split_sel].repeat(N, 1, 1))),
            "f_rest": torch.cat((self._features_rest[clone_sel], self._features_rest[split_sel].repeat(N, 1, 1))),
            "opacity": torch.cat((self._opacity[clone_sel], self._opacity[split_sel].repeat(N, 1))),
            "scaling": torch.cat((self._scaling[clone_sel], child_scaling)),
            "rotation": torch.cat((self._rotation[clone_sel], self._rotation[split_sel].repeat(N, 1))),
        }

        def pruned(opacity_raw, scaling_raw):
            m = (self.opacity_activation(opacity_raw) < min_opacity).squeeze(-1)
            if max_screen_size:     # max_radii2D was reset to zeros by the densification: only the world-size test can fire
                m = m | (torch.zeros_like(m, dtype=torch.float32) > max_screen_size) | \
                    (self.scaling_activation(scaling_raw).max(dim=1).values > max_world_size)
            return m
        keep_old = ~split_sel & ~pruned(self._opacity, self._scaling)
        keep_new = ~pruned(new_rows["opacity"], new_rows["scaling"])
        index = torch.cat((torch.nonzero(keep_old).squeeze(-1), P + torch.nonzero(keep_new).squeeze(-1))).contiguous()
        self._adopt(self._gather_all(index, P, new_rows))
        n = int(index.numel())
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)
        torch.cuda.empty_cache()

    def _densify_and_prune_stepwise(self, max_grad, min_opacity, extent, max_screen_size, max_world_size):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > max_world_size)
        self.prune_points(prune)
        if self.get_xyz.is_cuda:
            torch.cuda.empty_cache()

    def prune_only(self, min_opacity=0.05, max_world_size=0.01):
        prune = (self.get_opacity < min_opacity).squeeze() | (self.get_scaling.max(dim=1).values > max_world_size)
        self.prune_points(prune)
        if self.get_xyz.is_cuda:
            torch.cuda.empty_cache()

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """gaussian_model.py:420-422.  Written with a multiplicative mask instead of boolean-mask indexing: the same
        values (x + 0 == x), but no nonzero() and therefore no host synchronisation inside the training step."""
        m = update_filter.to(self.xyz_gradient_accum.dtype).unsqueeze(-1)
        self.xyz_gradient_accum += torch.norm(viewspace_point_tensor[:, :2], dim=-1, keepdim=True) * m
        self.denom += m
