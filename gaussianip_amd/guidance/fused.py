"""GroupNorm(+SiLU) as one fused HIP op for channels-last fp16 tensors (include/gip_nn.h, csrc/groupnorm.hip).

`GroupNormAct` is a drop-in nn.GroupNorm that optionally applies SiLU.  On a GPU, for fp16 channels-last inputs, it
runs the fused kernels through the C-ABI (forward: statistics pass + apply pass; backward: dL/dx only — the guidance
networks are frozen).  Any other input (CPU tests, fp32) takes the plain PyTorch ops with identical semantics.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_ws = {}


def _workspace(dev, nbytes):
    w = _ws.get(dev)
    if w is None or w.numel() < nbytes:
        w = _ws[dev] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    return w


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _FusedGN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, act):
        N, C, H, W = x.shape
        lib = _lib.nn_lib()
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mean = torch.empty((N, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        nb = lib.gip_gn_workspace_bytes(N, groups)
        ws = _workspace(x.device, nb)
        rc = lib.gip_gn_silu_forward(_p(x), _p(weight), _p(bias), _p(y), _p(mean), _p(rstd), N, H * W, C, groups,
                                     float(eps), int(act), _p(ws), ws.numel(),
                                     ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_gn_silu_forward failed with status %d" % rc)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.groups, ctx.act = groups, act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        lib = _lib.nn_lib()
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        nb = lib.gip_gn_workspace_bytes(N, ctx.groups)
        ws = _workspace(x.device, nb)
        rc = lib.gip_gn_silu_backward(_p(x), _p(dy), _p(weight), _p(bias), _p(mean), _p(rstd), _p(dx), N, H * W, C,
                                      ctx.groups, int(ctx.act), _p(ws), ws.numel(),
                                      ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_gn_silu_backward failed with status %d" % rc)
        return dx, None, None, None, None, None


class GroupNormAct(nn.GroupNorm):
    def __init__(self, num_groups, num_channels, eps=1e-5, act=False):
        super().__init__(num_groups, num_channels, eps=eps)
        self.act = act

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and x.shape[1] % 8 == 0 and
                x.is_contiguous(memory_format=torch.channels_last) and self.weight.dtype == torch.float16 and
                not self.weight.requires_grad):
            return _FusedGN.apply(x, self.weight, self.bias, self.num_groups, self.eps, self.act)
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.silu(y) if self.act else y
