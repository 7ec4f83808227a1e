"""LPIPS (VGG16) perceptual distance for the stage-3 reconstruction loss.

The reference builds `lpips.LPIPS(net='vgg')` (threestudio/systems/GaussianIP.py:121) and adds
`lambda_lpips * lpips(render_small, refined_small, normalize=True).mean()` to the L1 term (`:433-436`).  The `lpips` pip
package is a third-party dependency that is not vendored in the reference tree (imported at `GaussianIP.py:13`, absent
from `requirements.txt`, so no pinned version; the published algorithm is LPIPS v0.1, Zhang et al. 2018).  Parity
against the package itself is therefore UNPINNED (it is not installed here either); this file restates the published
algorithm:

    x -> 2x - 1 (normalize=True)  ->  (x - shift) / scale  ->  VGG16 conv stack, taps after relu1_2 / 2_2 / 3_3 / 4_3 / 5_3
    d = sum_l  mean_hw( lin_l( (f_l(a) / |f_l(a)|_c  -  f_l(b) / |f_l(b)|_c)^2 ) ),   lin_l = bias-free 1x1 convolution to 1 channel

Module and parameter names follow the package's state_dict (`net.slice{1..5}.{torchvision index}.weight`,
`lin{0..4}.model.1.weight`, `scaling_layer.shift / scale`), so `load_state_dict(torch.load(<lpips vgg checkpoint>),
strict=False)` after loading torchvision's `vgg16().features` weights into `net` fills every tensor.  No pretrained
weights ship here (no network): benchmarks and tests use `init_for_benchmark`.

On the GPU in fp16 / channels_last the twelve 64..512-channel 3x3 convolutions (and their data gradients towards the
rendered image) run on the MFMA implicit-GEMM kernel (csrc/conv3x3.hip) through `fused.conv3x3`; the 3 -> 64 stem uses
the zero-padded-input form (`fused.conv3x3_few_inputs`).  The distance itself is accumulated in fp32.  The features of
the fixed target images can be computed once and cached (`target_features` / `distance_to_features`): 32 refined views
at 415 x 290 are ~2.5 GB of fp16 features, nothing against 288 GB of HBM, and halve the per-step VGG work.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused

# torchvision vgg16().features indices of the convolutions inside each LPIPS slice (ReLU follows every convolution,
# MaxPool2d(2, 2) precedes slices 2-5)
_SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
_WIDTHS = (64, 128, 256, 512, 512)


class _Slice(nn.Module):
    def __init__(self, conv_ids, cin, cout, pool):
        super().__init__()
        self.pool = pool
        self.conv_ids = conv_ids
        for k, i in enumerate(conv_ids):
            self.add_module(str(i), nn.Conv2d(cin if k == 0 else cout, cout, 3, padding=1))

    def forward(self, x):
        if self.pool:
            x = F.max_pool2d(x, 2, 2)
        for i in self.conv_ids:
            conv = getattr(self, str(i))
            if conv.in_channels < 8:
                x = fused.conv3x3_few_inputs(x, conv.weight, conv.bias)
            else:
                x = fused.conv3x3(x, conv.weight, conv.bias)
            x = F.relu(x)
        return x


class _VGG16Features(nn.Module):
    def __init__(self):
        super().__init__()
        cin = 3
        for k, (ids, w) in enumerate(zip(_SLICES, _WIDTHS)):
            setattr(self, "slice%d" % (k + 1), _Slice(ids, cin, w, pool=k > 0))
            cin = w

    def forward(self, x):
        feats = []
        for k in range(5):
            x = getattr(self, "slice%d" % (k + 1))(x)
            feats.append(x)
        return feats


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class _Lin(nn.Module):
    """NetLinLayer: [Dropout (inactive: the package builds the model in eval mode), bias-free 1x1 convolution to 1 channel]."""

    def __init__(self, cin):
        super().__init__()
        self.model = nn.Sequential(nn.Identity(), nn.Conv2d(cin, 1, 1, bias=False))


class _HalfFeatures(torch.autograd.Function):
    """The fp16 VGG stack as ONE autograd node with fp32 inputs / outputs and a scaled backward.

    The distance is a mean over up to 120k positions of differences of unit vectors, so the gradients entering the
    feature maps are 1e-8 .. 1e-6 — below fp16's smallest subnormal (6e-8): fed to the fp16 graph directly they flush to
    zero (measured: 85 % of the image gradient's norm lost).  All five feature gradients arrive here together, so the
    node picks one power-of-two scale that puts their largest magnitude at 2^8 (device-side, no host sync), runs the
    inner fp16 backward on the scaled gradients and divides the fp32 image gradient by the same scale."""

    @staticmethod
    def forward(ctx, net, x):
        with torch.enable_grad():
            x16 = x.detach().to(torch.float16)
            if x16.is_cuda:
                x16 = x16.contiguous(memory_format=torch.channels_last)
            x16.requires_grad_(x.requires_grad)
            feats = net(x16)
        ctx.inner = (x16, feats)
        return tuple(f.detach().float() for f in feats)

    @staticmethod
    def backward(ctx, *grads):
        x16, feats = ctx.inner
        ctx.inner = None
        amax = torch.stack([g.abs().max() for g in grads]).max().clamp_min(1e-30)
        scale = torch.exp2(torch.floor(torch.log2(256.0 / amax))).clamp(1.0, 2.0 ** 60)
        g16 = [(g * scale).to(torch.float16) for g in grads]
        gx, = torch.autograd.grad(feats, x16, g16)
        return None, gx.float() / scale


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


class _DistanceToTargets(torch.autograd.Function):
    """x (fp32, already shifted / scaled) -> LPIPS distance [N, 1, 1, 1] to cached unit target features, as ONE node:
    fp16 VGG stack (MFMA convolutions), then per tap one fused kernel (csrc/lpips.hip) instead of ~12 fp32 elementwise /
    reduction passes; the backward writes the five feature gradients in fp16 with a power-of-two loss scale (see
    _HalfFeatures for why), runs the inner fp16 backward and unscales the fp32 image gradient."""

    @staticmethod
    def forward(ctx, module, x, *targets):
        from .. import _lib
        lib = _lib.nn_lib()
        needs_grad = x.requires_grad
        with torch.set_grad_enabled(needs_grad):
            x16 = _nhwc(x.detach().to(torch.float16)).requires_grad_(needs_grad)
            feats = [_nhwc(f) for f in module.net(x16)]
        N = x.shape[0]
        stream = fused.ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        lins = module.lin_vectors()
        targets = [_nhwc(t) for t in targets]
        total = torch.zeros(N, device=x.device, dtype=torch.float32)
        for f, t, lin in zip(feats, targets, lins):
            assert t.shape == f.shape and t.dtype == torch.float16, "target features do not match the rendered images"
            HW, C = f.shape[2] * f.shape[3], f.shape[1]
            blocks = lib.gip_lpips_layer_blocks(N, HW)
            partial = torch.empty(N, blocks, device=x.device, dtype=torch.float32)
            rc = lib.gip_lpips_layer_forward(fused._p(f), fused._p(t), fused._p(lin), fused._p(partial), N, HW, C, blocks, stream)
            if rc != 0:
                raise RuntimeError("gip_lpips_layer_forward failed with status %d" % rc)
            total += partial.sum(dim=1) / HW
        ctx.inner = (x16, feats, targets, lins) if needs_grad else None
        return total.view(N, 1, 1, 1)

    @staticmethod
    def backward(ctx, gout):
        from .. import _lib
        lib = _lib.nn_lib()
        x16, feats, targets, lins = ctx.inner
        ctx.inner = None
        N = x16.shape[0]
        g = gout.reshape(N).float()
        # |g_u| <= 2 lin |u - t| gout / HW <= 4 max(lin) gout / HW: put that bound at 2^8 (features of norm >= 1/256 stay finite;
        # the kernel saturates the rest)
        bound = torch.stack([4.0 * lin.max() / (f.shape[2] * f.shape[3]) for f, lin in zip(feats, lins)]).max() * g.abs().max()
        scale = torch.exp2(torch.floor(torch.log2(256.0 / bound.clamp_min(1e-30)))).clamp(1.0, 2.0 ** 60)
        stream = fused.ctypes.c_void_p(torch.cuda.current_stream(x16.device).cuda_stream)
        grads = []
        for f, t, lin in zip(feats, targets, lins):
            HW, C = f.shape[2] * f.shape[3], f.shape[1]
            coef = (g * (scale / HW)).contiguous()
            gf = torch.empty_like(f)
            rc = lib.gip_lpips_layer_backward(fused._p(f), fused._p(t), fused._p(lin), fused._p(coef), fused._p(gf), N, HW, C, stream)
            if rc != 0:
                raise RuntimeError("gip_lpips_layer_backward failed with status %d" % rc)
            grads.append(gf)
        gx, = torch.autograd.grad(feats, x16, grads)
        return (None, gx.float() / scale) + (None,) * len(targets)


def _unit(f, eps=1e-10):
    f = f.float()
    return f / (f.pow(2).sum(dim=1, keepdim=True).sqrt() + eps)


class LPIPSVGG(nn.Module):
    """`lpips.LPIPS(net='vgg')` forward (eval mode, spatial=False): returns the distance as [N, 1, 1, 1]."""

    def __init__(self):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.net = _VGG16Features()
        for k, w in enumerate(_WIDTHS):
            setattr(self, "lin%d" % k, _Lin(w))
        self.requires_grad_(False).eval()

    # ---- features ----
    def features(self, x, normalize=False):
        """The five channel-normalised feature maps (fp32) the distance is taken between."""
        if normalize:                      # [0, 1] -> [-1, 1]
            x = 2.0 * x - 1.0
        x = self.scaling_layer(x.float())
        if getattr(self.net.slice1, "0").weight.dtype == torch.float16:
            if x.requires_grad and torch.is_grad_enabled():
                raw = _HalfFeatures.apply(self.net, x)
            else:
                x = x.to(torch.float16)
                raw = self.net(x.contiguous(memory_format=torch.channels_last) if x.is_cuda else x)
        else:
            raw = self.net(x)
        return [_unit(f) for f in raw]

    @torch.no_grad()
    def target_features(self, x, normalize=False, dtype=torch.float16):
        """Features of fixed target images, for `distance_to_features` (stored in `dtype`)."""
        return [f.to(dtype) for f in self.features(x, normalize)]

    # ---- distance ----
    def _reduce(self, fa, fb):
        total = None
        for k in range(5):
            lin = getattr(self, "lin%d" % k).model[1].weight.float()          # [1, C, 1, 1]
            d = ((fa[k] - fb[k].float()) ** 2 * lin).sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)
            total = d if total is None else total + d
        return total

    def forward(self, in0, in1, normalize=False):
        return self._reduce(self.features(in0, normalize), self.features(in1, normalize))

    def distance_to_features(self, in0, target_feats, normalize=False):
        if (not fused._DISABLED and in0.is_cuda and getattr(self.net.slice1, "0").weight.dtype == torch.float16 and
                all(t.dtype == torch.float16 and t.is_cuda for t in target_feats)):
            x = self.scaling_layer((2.0 * in0 - 1.0 if normalize else in0).float())
            return _DistanceToTargets.apply(self, x, *target_feats)
        return self._reduce(self.features(in0, normalize), target_feats)

    def lin_vectors(self):
        """The five lin layers as contiguous fp32 [C] vectors (cached: the weights are frozen)."""
        w0 = self.lin0.model[1].weight
        if self._lin_cache is None or self._lin_cache[0].device != w0.device:
            self._lin_cache = [getattr(self, "lin%d" % k).model[1].weight.detach().float().reshape(-1).contiguous() for k in range(5)]
        return self._lin_cache

    _lin_cache = None

    # ---- weights ----
    def init_for_benchmark(self, seed=0):
        """He-initialised convolutions (activations of order one through the ReLU stack) and non-negative lin weights,
        as the published ones are: stands in for the pretrained weights, which cannot ship."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.net.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.weight[0].numel()
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                    m.bias.zero_()
            for k in range(5):
                w = getattr(self, "lin%d" % k).model[1].weight
                w.copy_(torch.rand(w.shape, generator=g) * (2.0 / w.shape[1]))
        self._lin_cache = None
        return self

    def prepare_inference(self, device=None, dtype=torch.float16):
        """Frozen fp16 / channels_last weights on the device: the layout the MFMA convolution reads directly."""
        self.to(device=device)
        self._lin_cache = None
        for m in self.net.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.to(dtype).contiguous(memory_format=torch.channels_last)
                m.bias.data = m.bias.data.to(dtype)
        return self
