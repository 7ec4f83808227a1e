"""torch.optim.Adam for the Gaussian model whose step() is ONE HIP launch over all parameter groups (include/gip_model.h,
gip_adam_step).  The reference builds six single-tensor groups with their own learning rates (gaussian_model.py:145-155);
torch's fused Adam then launches three multi-tensor kernels per group — 18 launches for 5.6 MB of state.  Same state layout
(state[p] = {"step", "exp_avg", "exp_avg_sq"}, the step count a device scalar like torch's fused / capturable Adam), so
the densify / prune surgery on the moments and `state_dict()` work unchanged; a GradScaler drives it like torch's fused
Adam (`_step_supports_amp_scaling`: found_inf stays on the device, a skipped step moves nothing)."""
import ctypes

import torch


class GipAdam(torch.optim.Adam):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self._step_supports_amp_scaling = True

    @staticmethod
    def _migrate_state(st, p):
        """State that came through load_state_dict() of a reference / plain torch.optim.Adam checkpoint: `step` is then a CPU
        tensor (or a Python number) and the moments may have another dtype / layout.  The kernel takes raw device pointers, so
        bring them into the form it walks: `step` a 0-d float32 tensor on p's device, moments float32 and dense in p's layout."""
        stp = st.get("step")
        if not (torch.is_tensor(stp) and stp.device == p.device and stp.dtype == torch.float32 and stp.dim() == 0):
            st["step"] = torch.as_tensor(float(stp) if stp is not None else 0.0, dtype=torch.float32).to(p.device).reshape(())
        for k in ("exp_avg", "exp_avg_sq"):
            m = st.get(k)
            if m is None:
                st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not (m.device == p.device and m.dtype == torch.float32 and m.shape == p.shape and m.stride() == p.stride()):
                if m.shape != p.shape:
                    raise ValueError("GipAdam: state %r has shape %s, parameter %s" % (k, tuple(m.shape), tuple(p.shape)))
                st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(m)

    @torch.no_grad()
    def step(self, closure=None):
        from .. import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        found_inf = getattr(self, "found_inf", None)
        grad_scale = getattr(self, "grad_scale", None)
        groups, keep = [], []
        betas = eps = None
        dev = None
        for g in self.param_groups:
            if betas is None:
                betas, eps = g["betas"], g["eps"]
            elif (betas, eps) != (g["betas"], g["eps"]):
                raise ValueError("GipAdam: all groups share betas / eps (the reference's six groups do)")
            if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("GipAdam: plain Adam only")
            for p in g["params"]:
                if p.grad is None or p.numel() == 0:          # (f_rest is [P, 0, 3] at the shipped sh_degree 0)
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.grad.device == p.device):
                    raise ValueError("GipAdam: float32 CUDA parameters and gradients only (group %r: %s %s %s, grad %s %s)" % (
                        g.get("name"), p.device, p.dtype, tuple(p.shape), p.grad.device, p.grad.dtype))
                permuted = False
                if not p.is_contiguous():
                    # a dense but permuted parameter (e.g. built from a transposed array): the flat kernel walks memory, so the
                    # gradient is brought into the parameter's own layout (zeros_like below gives the moments that layout too)
                    lay = torch.empty_like(p, memory_format=torch.preserve_format)
                    if lay.stride() != p.stride():
                        raise ValueError("GipAdam: parameter of group %r is not dense (strides %s)" % (g.get("name"), p.stride()))
                    p.grad = lay.copy_(p.grad)
                    permuted = True
                dev = p.device
                grad = p.grad if (permuted or p.grad.is_contiguous()) else p.grad.contiguous()
                if grad_scale is not None:              # scaler.step() without a preceding unscale_(): unscale here
                    grad = grad / grad_scale.to(grad.dtype)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                else:
                    self._migrate_state(st, p)
                keep.append(grad)
                groups.append(_lib.GipAdamGroup(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                st["step"].data_ptr(), p.numel(), float(g["lr"]), 0))
        if not groups:
            return loss
        lib = _lib.model_lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        fi = ctypes.c_void_p(None)
        if found_inf is not None:
            fi32 = found_inf.to(device=dev, dtype=torch.float32)       # (a converted temporary must outlive the launch)
            keep.append(fi32)
            fi = ctypes.c_void_p(fi32.data_ptr())
        for i in range(0, len(groups), 8):
            chunk = groups[i:i + 8]
            arr = (_lib.GipAdamGroup * len(chunk))(*chunk)
            rc = lib.gip_adam_step(arr, len(chunk), float(betas[0]), float(betas[1]), float(eps), fi, stream)
            if rc != 0:
                raise RuntimeError("gip_adam_step failed with status %d" % rc)
        return loss
