"""Fused HIP ops for the channels-last fp16 denoiser / VAE (include/gip_nn.h, csrc/groupnorm.hip).

`GroupNormAct` is a drop-in nn.GroupNorm that optionally applies SiLU and optionally adds a per-(sample, channel)
vector to its input on load (conv bias + time-embedding projection of ResnetBlock2D).  On a GPU, for fp16
channels-last inputs, it runs the fused kernels through the C-ABI (forward: statistics pass + apply pass; backward:
dL/dx only — the guidance networks are frozen).  `add_bias_residual` (shortcut + conv2 + biases) and `geglu` are the
pointwise companions.  Any other input (CPU tests, fp32) takes the plain PyTorch ops with identical semantics.
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_ws = {}


def _workspace(dev, nbytes):
    """Scratch of the split-K / GroupNorm kernels, one per (device, stream): work enqueued on different streams (the
    ControlNet runs beside the U-Net encoder) must not share it."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        with torch.cuda.stream(torch.cuda.current_stream(dev)):
            w = _ws[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    return w


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _FusedGN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, act, addend, chan_stats=None):
        N, C, H, W = x.shape
        ad_ptr, ad_stride = ctypes.c_void_p(None), 0
        if addend is not None:
            ad_ptr, ad_stride = _p(addend), (addend.stride(0) if addend.dim() == 2 and addend.shape[0] > 1 else 0)
        lib = _lib.nn_lib()
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mean = torch.empty((N, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if chan_stats is not None:
            # the producer of x (MFMA convolution / linear epilogue) already summed x per 128-row block and channel
            rc = lib.gip_gn_silu_forward_stats(_p(x), _p(weight), _p(bias), _p(y), _p(mean), _p(rstd), N, H * W, C, groups,
                                               float(eps), int(act), ad_ptr, ad_stride, _p(chan_stats), chan_stats.shape[0] // N, stream)
        else:
            nb = lib.gip_gn_workspace_bytes(N, groups)
            ws = _workspace(x.device, nb)
            rc = lib.gip_gn_silu_forward(_p(x), _p(weight), _p(bias), _p(y), _p(mean), _p(rstd), N, H * W, C, groups,
                                         float(eps), int(act), ad_ptr, ad_stride, _p(ws), ws.numel(), stream)
        if rc != 0:
            raise RuntimeError("gip_gn_silu_forward failed with status %d" % rc)
        ctx.save_for_backward(x, weight, bias, mean, rstd, addend)
        ctx.groups, ctx.act, ctx.ad_stride = groups, act, ad_stride
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, rstd, addend = ctx.saved_tensors
        ad_ptr = ctypes.c_void_p(None) if addend is None else _p(addend)
        N, C, H, W = x.shape
        lib = _lib.nn_lib()
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        nb = lib.gip_gn_workspace_bytes(N, ctx.groups)
        ws = _workspace(x.device, nb)
        rc = lib.gip_gn_silu_backward(_p(x), _p(dy), _p(weight), _p(bias), _p(mean), _p(rstd), _p(dx), N, H * W, C,
                                      ctx.groups, int(ctx.act), ad_ptr, ctx.ad_stride, _p(ws), ws.numel(),
                                      ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_gn_silu_backward failed with status %d" % rc)
        return dx, None, None, None, None, None, None, None


class GroupNormAct(nn.GroupNorm):
    def __init__(self, num_groups, num_channels, eps=1e-5, act=False):
        super().__init__(num_groups, num_channels, eps=eps)
        self.act = act

    def forward(self, x, addend=None):
        """`addend` ([N, C], [1, C] or [C], unit stride on C) is added to x before the normalisation."""
        if fusable(x) and self.weight.dtype == torch.float16 and not self.weight.requires_grad and \
                (addend is None or (addend.dtype == torch.float16 and addend.stride(-1) == 1 and not addend.requires_grad)):
            return _FusedGN.apply(x, self.weight, self.bias, self.num_groups, self.eps, self.act, addend, producer_stats(x))
        fallback("GroupNormAct", x)
        if addend is not None:
            x = x + addend.reshape(-1 if addend.dim() == 2 and addend.shape[0] > 1 else 1, x.shape[1], 1, 1)
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.silu(y) if self.act else y


_DISABLED = False
_STATS_ATTR = "_gip_chan_stats"

# ---- GIP_STRICT: no silent fallbacks -----------------------------------------------------------------------------------------------
# Every place where an fp16 GPU tensor of the networks leaves this repo's HIP kernels calls `fallback(site, tensor)` first.
#   GIP_STRICT=1  a SHAPE fallback (a layer the kernels do not take: F.conv2d / F.group_norm / SDPA / F.linear "anything else")
#                 raises instead of running on MIOpen / AOTriton / hipBLASLt unnoticed — round 4's irreproducible sharded denoise
#                 (20-30-tile convolutions silently on MIOpen's atomic split-K) is what this catches on day one;
#   GIP_STRICT=2  the vendor-library calls that are there BY DESIGN (library=True: measured dispatch to hipBLASLt, MIOpen
#                 backward-data of two tiny layers) raise as well — the bar for "no vendor library on the hot path".
# `fallback_counts` counts every such call whatever the level (tests / tools read it).
fallback_counts = {}


_tuned_gemms = None


def enable_tuned_gemms():
    """The hipBLASLt solution per library-GEMM shape of the step, picked once by PyTorch's TunableOp on an MI355X box and shipped as a
    results file (guidance/tunableop_gfx950.csv: 20 half-precision shapes — the Winograd products, ff_in / q|k|v below 64^2, the
    prompt-token projections, the VAE's dense attention products; hipBLASLt solutions only, rocBLAS candidates dropped).  Reading it
    switches TunableOp ON with tuning OFF: a listed shape runs its tuned solution, everything else the library's default; a file
    taken with other library versions fails TunableOp's validators and is ignored.  Same-box A/B of the AHDS step: 33.84 / 33.65 ms
    default, 33.56 / 33.49 tuned.  GIP_TUNABLEOP=0 switches it off.  Returns whether the tuned table is active."""
    global _tuned_gemms
    if _tuned_gemms is not None:
        return _tuned_gemms
    _tuned_gemms = False
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
    if os.environ.get("GIP_TUNABLEOP", "1") == "0" or not os.path.exists(path) or not torch.cuda.is_available():
        return False
    try:
        import torch.cuda.tunable as tun
        if tun.is_enabled():
            return False                       # the host application drives TunableOp itself: leave its settings alone
        tun.enable(True)
        tun.tuning_enable(False)
        tun.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "gip_tunableop_unused.csv"))    # never write next to the sources
        _tuned_gemms = bool(tun.read_file(path))
        if not _tuned_gemms:
            tun.enable(False)
    except Exception:      # noqa: BLE001  (an older PyTorch without the module: the library's defaults)
        _tuned_gemms = False
    return _tuned_gemms


def fallback(site, t, library=False):
    if _DISABLED or not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float16):
        return
    fallback_counts[site] = fallback_counts.get(site, 0) + 1
    level = int(os.environ.get("GIP_STRICT", "0") or 0)
    if level >= (2 if library else 1):
        raise RuntimeError("GIP_STRICT=%d: %s left the HIP kernels for a %s (tensor %s)" % (
            level, site, "vendor-library call" if library else "PyTorch / vendor-library FALLBACK", tuple(t.shape)))

# Captured HIP graphs (ipa_guidance._forward_unet_graph / _encode_graphed) freeze what they saw at capture: weight values
# are read in place, but DERIVED copies (Winograd U, q|k|v concatenations, packed time-embedding / context projections),
# scalar kernel arguments (Attention.ip_scale) and the A/B environment switches are baked in.  Everything that changes one
# of those bumps this counter; it is part of the graph keys, so a stale graph is never replayed.
_weights_epoch = 0
_ENV_KNOBS = ("GIP_WINOGRAD", "GIP_WINOGRAD_SHAPES", "GIP_WINOGRAD_GEMM", "GIP_GN_STATS", "GIP_FUSE_QKV", "GIP_GN_BWD_SUMS", "GIP_RESBLOCK_NODE",
              "GIP_OWN_GEMM", "GIP_CONV_HALO", "GIP_CONV_GNIN", "GIP_WINOGRAD_GN", "GIP_TUNABLEOP", "GIP_LN_FOLD", "GIP_CONV_S2_STATS")


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1
    return _weights_epoch


def graph_signature():
    """What a captured graph depends on besides its input shapes: the weights epoch and the A/B switches."""
    return (_weights_epoch, bool(_DISABLED)) + tuple(os.environ.get(k) for k in _ENV_KNOBS)


def producer_stats(x):
    """chan_stats [N * HW / R, C, 2] float32 (R = 128 pixels per block — 128 consecutive pixels, or one 16 x 8 image block
    when the halo-resident convolution kernel produced x; a block never straddles two samples and the consumers only sum a
    sample's blocks — or 64 at the 8 x 8 level) that the kernel which produced `x` wrote in its epilogue (attached to the tensor OBJECT by conv3x3 / linear below; any view / copy of x drops
    it and the GroupNorm takes its own statistics pass), or None."""
    st = getattr(x, _STATS_ATTR, None)
    if st is None or x.dim() != 4:
        return None
    N, C, H, W = x.shape
    R = (N * H * W) // max(int(st.shape[0]), 1)
    if R not in (64, 128) or st.shape != (N * H * W // R, C, 2) or (H * W) % R or C // 32 > 256 or os.environ.get("GIP_GN_STATS", "1") == "0":
        return None
    return st


def attach_stats(x, st):
    if st is not None:
        setattr(x, _STATS_ATTR, st)
    return x


def stats_wanted(N, H, W, cout):
    """The producer takes the next GroupNorm's statistics when a 128-pixel block never straddles two samples (whole-K
    tiles write them in their epilogue, split-K layers in their reduce kernel)."""
    return (not _DISABLED and (H * W) % 128 == 0 and cout % 8 == 0 and
            _conv_tiles(N, H, W, cout) >= 256 and
            os.environ.get("GIP_GN_STATS", "1") != "0")


def _stats_rows(N, H, W, cin, cout):
    """Rows per statistics block the 3x3 convolution delivers for this shape: 128 or 0 (none).  (Statistics out of the split-K
    reduce kernel, incl. 64-row blocks at the 8 x 8 level, measured neutral in round 4 and were removed in round 5.)"""
    return 128 if stats_wanted(N, H, W, cout) else 0


class disabled:
    """Context manager: route everything through the plain PyTorch ops (used to COUNT algorithmic FLOPs with
    torch.utils.flop_counter, which cannot see the ctypes-launched HIP kernels, and by A/B tests)."""

    def __enter__(self):
        global _DISABLED
        self._old, _DISABLED = _DISABLED, True

    def __exit__(self, *exc):
        global _DISABLED
        _DISABLED = self._old


def fusable(x):
    """fp16 NHWC activations on a GPU: the layout / dtype the HIP kernels are written for."""
    return (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and x.shape[1] % 8 == 0 and
            x.is_contiguous(memory_format=torch.channels_last))


class _AddBiasResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bias):
        N, C, H, W = a.shape
        out = torch.empty_like(a, memory_format=torch.channels_last)
        rc = _lib.nn_lib().gip_add_bias_residual(_p(a), _p(b), ctypes.c_void_p(None) if bias is None else _p(bias), _p(out),
                                                 N * H * W, C, ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_add_bias_residual failed with status %d" % rc)
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, None


def add_bias_residual(a, b, bias=None):
    """a + b + bias[None, :, None, None] in one pass (biases are frozen: no gradient for them)."""
    if fusable(a) and fusable(b) and a.shape == b.shape and (bias is None or (bias.dtype == torch.float16 and not bias.requires_grad)):
        return _AddBiasResidual.apply(a, b, bias)
    out = a + b
    return out if bias is None else out + bias.reshape(1, -1, 1, 1)


def cat_skip(h, skip, residual=None):
    """torch.cat([h, skip (+ residual)], dim=1) of a U-Net up-block layer in ONE pass that also takes the statistics of the
    ResnetBlock2D.norm1 that consumes it (attached to the result, see producer_stats) — instead of an add kernel, a cat
    kernel and the GroupNorm's own statistics read.  `skip + residual` is rounded to half before the concatenation, as the
    separate add rounds it."""
    if (fusable(h) and fusable(skip) and (residual is None or (fusable(residual) and residual.shape == skip.shape)) and
            h.shape[0] == skip.shape[0] and h.shape[2:] == skip.shape[2:] and h.shape[1] % 64 == 0 and skip.shape[1] % 64 == 0 and
            not (torch.is_grad_enabled() and (h.requires_grad or skip.requires_grad or (residual is not None and residual.requires_grad)))):
        N, Ca, H, W = h.shape
        Cb = skip.shape[1]
        out = torch.empty((N, Ca + Cb, H, W), dtype=h.dtype, device=h.device, memory_format=torch.channels_last)
        want = (H * W) % 128 == 0 and (Ca + Cb) // 32 <= 256 and os.environ.get("GIP_GN_STATS", "1") != "0"
        st = torch.empty((N * H * W // 128, Ca + Cb, 2), dtype=torch.float32, device=h.device) if want else None
        null = ctypes.c_void_p(None)
        rc = _lib.nn_lib().gip_cat2_stats_f16(_p(h), _p(skip), null if residual is None else _p(residual), _p(out),
                                              null if st is None else _p(st), N * H * W, Ca, Cb,
                                              ctypes.c_void_p(torch.cuda.current_stream(h.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_cat2_stats_f16 failed with status %d" % rc)
        return attach_stats(out, st)
    return torch.cat([h, skip if residual is None else skip + residual], dim=1)


def geglu(x):
    """diffusers GEGLU on the projected tensor: value, gate = x.chunk(2, -1); value * gelu(gate)."""
    D = x.shape[-1] // 2
    if not _DISABLED and x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and D % 8 == 0 and not (x.requires_grad and torch.is_grad_enabled()):
        out = torch.empty(x.shape[:-1] + (D,), dtype=x.dtype, device=x.device)
        rc = _lib.nn_lib().gip_geglu(_p(x), _p(out), x.numel() // (2 * D), D,
                                     ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_geglu failed with status %d" % rc)
        return out
    a, g = x.chunk(2, dim=-1)
    return a * F.gelu(g)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm over the last dimension with the same parameters and state_dict keys; frozen fp16 inference on the
    GPU runs csrc/groupnorm.hip's one-read-one-write row kernel (BasicTransformerBlock.norm1 / norm2 / norm3)."""

    def forward(self, x):
        C = x.shape[-1]
        if (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and self.weight is not None and
                self.bias is not None and self.weight.dtype == torch.float16 and len(self.normalized_shape) == 1 and
                C % 8 == 0 and C <= 2048 and x.numel() > 0 and
                not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad))):
            out = torch.empty_like(x)
            rc = _lib.nn_lib().gip_layernorm_f16(_p(x), _p(self.weight), _p(self.bias), _p(out), x.numel() // C, C,
                                                 float(self.eps), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("gip_layernorm_f16 failed with status %d" % rc)
            return out
        fallback("LayerNorm", x)
        return super().forward(x)


# ---------------------------------------------------------------------------------------------------------------------
# 3x3 / stride 1 / pad 1 convolution on the matrix cores (csrc/conv3x3.hip)
# ---------------------------------------------------------------------------------------------------------------------
_GN_SUMS_MIN_TILES = 256    # below it the data gradient runs split-K, whose reduce kernel does not make the sums
# Tile counts below 256 run split-K (fp32 slabs in the shared workspace).  The floor was 32 until round 4: below it the layer
# went to MIOpen — which is what the 8 x 8 level of a 2-view shard (batch 6: 30 tiles) and the 16 x 16 / 8 x 8 levels of a
# 1-view shard (batch 3) did, and MIOpen's small-problem kernels there are atomic split-K ones: the denoise of a sharded
# step was NOT reproducible run to run (tools/diag/denoise_bisect.py: first differing module down_sample.2 at batch 6,
# down_sample.1 at batch 3; 2.4e-3 of the output) and slow.  The MFMA kernel with split-K 16 takes them all (bitwise
# reproducible: fixed summation order); only single-tile problems stay on the library.
_MIN_CONV_TILES = 2
_UPCONV_MIN_TILES = 256      # smallest grid (128-row tiles over the four parity classes) upsample + convolution takes as one launch
_SPLITK_WS_BYTES = 64 << 20
_WT_CACHE_MAX = 512


class _WeightCache:
    """Derived (flipped / transposed) copies of frozen convolution weights, keyed by (tag, data_ptr, version, shape).
    Every entry keeps a STRONG reference to the source tensor: while the entry lives the source's storage cannot be freed,
    so the allocator cannot hand the same address (at version 0, same shape) to a different weight and make the entry
    stale.  Least-recently-used entries beyond _WT_CACHE_MAX are dropped (a rebuilt network does not pin the old one's
    weights forever) — EXCEPT entries that were created or read while a HIP graph was being captured: the graph bakes
    their device addresses into its kernel arguments, so they stay (pinned, per owner of the capturing graph: capture_owner)
    until the graphs that use them are dropped (`unpin(owner)`, called by StableDiffusionGuidance.invalidate_graphs)."""

    def __init__(self):
        from collections import OrderedDict
        self._d = OrderedDict()
        self._pinned = {}           # key -> set of owner tokens whose captured graphs use the entry
        self.owner = None           # token of whoever is capturing right now (capture_owner below)

    def _pin(self, key):
        self._pinned.setdefault(key, set()).add(self.owner)

    def get(self, tag, w, make):
        key = (tag, w.data_ptr(), w._version, tuple(w.shape), w.dtype)
        capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
        hit = self._d.get(key)
        if hit is not None:
            self._d.move_to_end(key)
            if capturing:
                self._pin(key)
            return hit[1]
        wt = make(w)         # (inside a capture: the copy kernels become part of the graph and the tensor lives in its pool — still correct)
        self._d[key] = (w, wt)
        if capturing:
            self._pin(key)
        # the budget counts UNPINNED entries only (round 6): with several guidance instances alive (a pytest process) the entries their
        # graphs pin used to eat the whole budget, so that what the eager warm-up call of a NEW instance derived was evicted again
        # before its capturing call — which then derived it inside the capture (and the Winograd transform's constants could not
        # be uploaded there: hipErrorStreamCaptureUnsupported)
        loose = [k for k in self._d if k not in self._pinned]
        for k in loose[:max(0, len(loose) - _WT_CACHE_MAX)]:
            del self._d[k]
        return wt

    def __len__(self):
        return len(self._d)

    def pinned(self):
        return len(self._pinned)

    def unpin(self, owner):
        """Release the pins of ONE owner's graphs (another guidance instance's live graphs keep theirs)."""
        for k in [k for k, owners in self._pinned.items() if owner in owners]:
            self._pinned[k].discard(owner)
            if not self._pinned[k]:
                del self._pinned[k]

    def unpin_all(self):
        self._pinned.clear()

    def clear(self):
        self._d.clear()
        self._pinned.clear()


class capture_owner:
    """`with capture_owner(token):` — derived weights created or read by a HIP-graph capture inside are pinned on behalf of
    `token` (the object that owns the graph) and released by `_wt_cache.unpin(token)` when it drops its graphs."""

    def __init__(self, token):
        self.token = token

    def __enter__(self):
        self._old, _wt_cache.owner = _wt_cache.owner, self.token

    def __exit__(self, *exc):
        _wt_cache.owner = self._old


_wt_cache = _WeightCache()


def _conv_tiles(N, H, W, cout):
    bn = 160 if (cout % 160 == 0 and cout % 128 != 0) else 128
    return ((N * H * W + 127) // 128) * ((cout + bn - 1) // bn)


def _conv_call(x, w, cout, bias=None, residual=None, stats=None):
    """`stats`: a list that receives the output's chan_stats tensor (see producer_stats) when the shape qualifies."""
    N, C, H, W = x.shape
    out = torch.empty((N, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    null = ctypes.c_void_p(None)
    rows = _stats_rows(N, H, W, C, cout) if stats is not None else 0
    if rows:
        st = torch.empty((N * H * W // rows, cout, 2), dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, _SPLITK_WS_BYTES) if _conv_tiles(N, H, W, cout) < 256 else None
        rc = _lib.nn_lib().gip_conv3x3_stats_ws_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias),
                                                         null if residual is None else _p(residual), _p(out), N, H, W, C, cout,
                                                         _p(st), rows, null if ws is None else _p(ws), 0 if ws is None else ws.numel(),
                                                         ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_conv3x3_stats_ws_nhwc_f16 failed with status %d" % rc)
        stats.append(st)
        return out
    ws = _workspace(x.device, _SPLITK_WS_BYTES) if _conv_tiles(N, H, W, cout) < 256 else None
    rc = _lib.nn_lib().gip_conv3x3_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias),
                                            null if residual is None else _p(residual), _p(out), N, H, W, C, cout,
                                            null if ws is None else _p(ws), 0 if ws is None else ws.numel(),
                                            ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_conv3x3_nhwc_f16 failed with status %d" % rc)
    return out


def _transposed_weight(w):
    """Weight of the data-gradient convolution: w_t[ci][2-dy][2-dx][co] = w[co][dy][dx][ci] (frozen weights: cached)."""
    return _wt_cache.get("t", w, lambda t: t.detach().flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last))


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, residual, stats=None):
        ctx.save_for_backward(w)
        ctx.x_shape, ctx.has_res = tuple(x.shape), residual is not None
        return _conv_call(x, w, w.shape[0], bias, residual, stats)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        N, _, H, W = ctx.x_shape
        if w.shape[0] % 64 == 0 and _conv_tiles(N, H, W, w.shape[1]) >= _MIN_CONV_TILES:
            dx = _conv_call(dy, _transposed_weight(w), w.shape[1])
        else:
            fallback("conv3x3 data gradient", dy)
            dx = torch.nn.grad.conv2d_input(ctx.x_shape, w, dy, padding=1)
        return dx, None, None, (dy if ctx.has_res else None), None


_WINOGRAD_G = {}


def _winograd_weight(w):
    """U [16][Cout][Cin] = (G g G^T)[i][j] of the 3x3 weight, fp32 arithmetic, one rounding to half."""
    G = _WINOGRAD_G.get(w.device)          # per device, uploaded once: a derivation inside a graph capture must not copy from the host
    if G is None:
        G = _WINOGRAD_G[w.device] = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float32, device=w.device)
    g = w.detach().float()                                     # [co, ci, 3, 3]
    U = torch.einsum("ik,ockl,jl->ijoc", G, g, G)             # [4, 4, co, ci]
    return U.reshape(16, w.shape[0], w.shape[1]).to(w.dtype).contiguous()


# grid size : input channels : fewest pixels (N * H * W), measured faster than the implicit GEMM (tools/diag/winograd_shapes.py at
# 12 samples, tools/diag/winograd_small_batch.py at the 3 / 6 samples of a sharded step): the 16 x 16 level and the wide
# skip-concatenation inputs of the 32 x 32 level; 640 / 320 channels at 32 x 32, everything at 64 x 64 (transform traffic) and
# at 8 x 8 (weight-bound: the transformed weights are 16 / 9 as large) stay on the implicit GEMM
_WINOGRAD_DEFAULT = "16:640:3072,16:1280:1536,16:1920:1536,16:2560:1536,32:960:3072,32:1280:3072,32:1920:3072"


# "own": the sixteen Winograd products on this repo's batched MFMA GEMM; "lib": one batched hipBLASLt call (torch.bmm); "auto" (round 6):
# the own GEMM where it measures at or below the library's time under graph replay — its 256 x 256 tiles (one 8-wave workgroup per
# CU; Cout % 256 == 0: the 16 x 16 level) at K <= 1280: 51.6 vs 52.3 us (1280 -> 1280, batch 12), 31.7 vs 32.0 (640 -> 1280), 0.77-0.99 at
# batch 3 / 6; the library's stream-K tiles win at K >= 1920 (1.11-1.14x) and on the 640-wide products of the 32 x 32 level.
# Same-box A/B: tools/exp_winograd_gemm.py (profiles/r06_winograd_gemm_256.txt, profiles/r05_winograd_gemm_batched.txt)
_WINOGRAD_GEMM_DEFAULT = "auto"


def _winograd_gemm_own(C, cout):
    mode = os.environ.get("GIP_WINOGRAD_GEMM", _WINOGRAD_GEMM_DEFAULT)
    return mode == "own" or (mode == "auto" and cout % 256 == 0 and C <= 1280)


def _winograd_shapes():
    """{(H, Cin): fewest pixels} the Winograd path takes; GIP_WINOGRAD_SHAPES="H:Cin[:pixels],..." overrides the measured
    default (experiments)."""
    out = {}
    for item in os.environ.get("GIP_WINOGRAD_SHAPES", _WINOGRAD_DEFAULT).split(","):
        if item:
            v = [int(t) for t in item.split(":")]
            out[(v[0], v[1])] = v[2] if len(v) > 2 else 0
    return out


def _winograd_applies(x, w, residual):
    """Where F(2x2, 3x3) beats the implicit GEMM (_WINOGRAD_DEFAULT above).  Frozen weights, no gradient path."""
    N, C, H, W = x.shape
    return (os.environ.get("GIP_WINOGRAD", "1") != "0" and fusable(x) and w.dtype == torch.float16 and not w.requires_grad and
            tuple(w.shape[2:]) == (3, 3) and w.shape[0] % 8 == 0 and H == W and N * H * W >= _winograd_shapes().get((H, C), 1 << 62) and
            not (torch.is_grad_enabled() and x.requires_grad) and (residual is None or fusable(residual)))


def _winograd_conv(x, w, bias, residual, stats=None, gn_in=None):
    """`stats`: a list that receives the output's chan_stats (see producer_stats) — the output transform takes them.
    `gn_in` = (GroupNormAct, addend, chan_stats of x): x is the RAW input of that GroupNorm (+ SiLU), which the input transform
    applies while it loads the patches (gip_winograd_input_gn_f16) — no apply pass, no normalised tensor."""
    N, C, H, W = x.shape
    cout = w.shape[0]
    T = N * (H // 2) * (W // 2)
    lib = _lib.nn_lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    V = torch.empty((16, T, C), dtype=x.dtype, device=x.device)
    if gn_in is None:
        rc = lib.gip_winograd_input_f16(_p(x), _p(V), N, H, W, C, stream)
    else:
        gn, addend, chan_stats = gn_in
        ad_ptr, ad_stride = ctypes.c_void_p(None), 0
        if addend is not None:
            ad_ptr, ad_stride = _p(addend), (addend.stride(0) if addend.dim() == 2 and addend.shape[0] > 1 else 0)
        mean = torch.empty((N, gn.num_groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        rc = lib.gip_gn_stats_from_partials(_p(mean), _p(rstd), N, H * W, C, gn.num_groups, float(gn.eps), ad_ptr, ad_stride, _p(chan_stats),
                                            chan_stats.shape[0] // N, stream)
        if rc != 0:
            raise RuntimeError("gip_gn_stats_from_partials failed with status %d" % rc)
        rc = lib.gip_winograd_input_gn_f16(_p(x), _p(V), N, H, W, C, _p(gn.weight), _p(gn.bias), _p(mean), _p(rstd), gn.num_groups, int(gn.act),
                                           ad_ptr, ad_stride, stream)
    if rc != 0:
        raise RuntimeError("gip_winograd_input_f16 failed with status %d" % rc)
    U = _wt_cache.get("wino", w, _winograd_weight)
    if _winograd_gemm_own(C, cout) and C % 64 == 0 and cout % 4 == 0 and T * max(C, cout) * 2 < (1 << 31):
        # the sixteen products in ONE launch of this repo's MFMA GEMM (blockIdx.y = product): gip_linear_batched_f16
        M = torch.empty((16, T, cout), dtype=x.dtype, device=x.device)
        rc = lib.gip_linear_batched_f16(_p(V), _p(U), _p(M), 16, T, C, cout, T * C, cout * C, T * cout, stream)
        if rc != 0:
            raise RuntimeError("gip_linear_batched_f16 failed with status %d" % rc)
    else:
        fallback("winograd batched GEMM", V, library=True)
        M = torch.bmm(V, U.transpose(1, 2))          # sixteen GEMMs: one batched library call
    out = torch.empty((N, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    null = ctypes.c_void_p(None)
    if W in (16, 32) and (H * W) % 128 == 0:
        st = None
        if stats is not None and cout // 32 <= 256 and os.environ.get("GIP_GN_STATS", "1") != "0":
            st = torch.empty((N * H * W // 128, cout, 2), dtype=torch.float32, device=x.device)
            stats.append(st)
        rc = lib.gip_winograd_output_stats_f16(_p(M), null if bias is None else _p(bias), null if residual is None else _p(residual),
                                               _p(out), null if st is None else _p(st), N, H, W, cout, stream)
    else:
        rc = lib.gip_winograd_output_f16(_p(M), null if bias is None else _p(bias), null if residual is None else _p(residual), _p(out),
                                         N, H, W, cout, stream)
    if rc != 0:
        raise RuntimeError("gip_winograd_output_f16 failed with status %d" % rc)
    return out


def conv3x3(x, w, bias=None, residual=None, gn_next=False):
    """F.conv2d(x, w, bias, padding=1) (+ residual) for a 3x3 kernel.  fp16 NHWC activations with Cin % 64 == 0 and
    enough output tiles to fill the chip run on the hand-written MFMA implicit GEMM with bias / residual in its
    epilogue; everything else goes to MIOpen.  `gn_next`: the result feeds a GroupNorm — the kernel's epilogue then also
    takes that GroupNorm's per-channel sums (attached to the returned tensor, see producer_stats)."""
    if _winograd_applies(x, w, residual) and (bias is None or not bias.requires_grad):
        if not gn_next:
            return _winograd_conv(x, w, bias, residual)
        holder = []
        out = _winograd_conv(x, w, bias, residual, holder)
        return attach_stats(out, holder[0] if holder else None)
    if (fusable(x) and x.shape[1] % 64 == 0 and w.shape[0] % 4 == 0 and w.dtype == torch.float16 and
            not w.requires_grad and w.is_contiguous(memory_format=torch.channels_last) and
            (bias is None or not bias.requires_grad) and (residual is None or fusable(residual)) and
            _conv_tiles(x.shape[0], x.shape[2], x.shape[3], w.shape[0]) >= _MIN_CONV_TILES and
            x.numel() * 2 < (1 << 31) and x.shape[0] * x.shape[2] * x.shape[3] * w.shape[0] * 2 < (1 << 31)):
        if not gn_next:
            return _Conv3x3.apply(x, w, bias, residual)
        holder = []
        out = _Conv3x3.apply(x, w, bias, residual, holder)
        return attach_stats(out, holder[0] if holder else None)
    fallback("conv3x3", x)
    out = F.conv2d(x, w, None, padding=1)
    if residual is not None:
        return add_bias_residual(residual, out, bias)
    return out if bias is None else out + bias.reshape(1, -1, 1, 1)


def conv3x3_gn(x, gn, addend, w, bias=None, residual=None, gn_next=False):
    """conv3x3(gn(x, addend), w, bias, residual, gn_next) — ResnetBlock2D's conv(silu(norm(x))).  Where the convolution runs as
    Winograd F(2x2, 3x3) and x carries its producer's per-channel sums, the GroupNorm (+ SiLU) happens inside the input transform:
    one pass over x instead of apply (read + write) + transform (read).  GIP_WINOGRAD_GN=0 switches it off (same-box A/B)."""
    st = producer_stats(x) if fusable(x) else None
    if (st is not None and os.environ.get("GIP_WINOGRAD_GN", "1") != "0" and _winograd_applies(x, w, residual) and
            (bias is None or not bias.requires_grad) and gn.weight.dtype == torch.float16 and not gn.weight.requires_grad and
            x.shape[1] % gn.num_groups == 0 and
            (addend is None or (addend.dtype == torch.float16 and addend.stride(-1) == 1 and not addend.requires_grad))):
        holder = [] if gn_next else None
        out = _winograd_conv(x, w, bias, residual, holder, gn_in=(gn, addend, st))
        return attach_stats(out, holder[0] if holder else None)
    return conv3x3(gn(x, addend), w, bias, residual, gn_next)


def conv1x1(x, w, bias=None):
    """1x1 convolution.  For NHWC fp16 activations it IS a dense GEMM on the [N*H*W, Cin] view: one hipBLASLt call
    with the bias in its epilogue (the library path for plain GEMMs) instead of MIOpen's conv + fill + bias kernels."""
    if fusable(x) and w.is_contiguous(memory_format=torch.channels_last):
        N, C, H, W = x.shape
        y = linear_auto(x.permute(0, 2, 3, 1).reshape(N * H * W, C), w.reshape(w.shape[0], C), bias)
        return y.view(N, H, W, w.shape[0]).permute(0, 3, 1, 2)
    fallback("conv1x1", x)
    return F.conv2d(x, w, bias)


def _kv_rows(t):
    """Row stride (in halves) of a [B, N, C] key / value tensor the attention kernel can read in place: packed, or a column
    range of a wider row-major matrix (rows of one sample consecutive).  None when a copy is needed."""
    B, N, C = t.shape
    if t.stride(2) != 1 or t.stride(1) % 8 or t.stride(1) < C or (B > 1 and t.stride(0) != N * t.stride(1)) or t.data_ptr() % 16:
        return None
    return t.stride(1)


def qkv_fusable(x, wq):
    return (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and not wq.requires_grad and wq.dtype == torch.float16 and
            not (torch.is_grad_enabled() and x.requires_grad) and os.environ.get("GIP_FUSE_QKV", "1") != "0")


def qkv_weight(wq, wk, wv):
    """[3C, C] concatenation of a self-attention layer's frozen to_q / to_k / to_v weights (cached like the transposed
    convolution weights: keyed by the q weight's storage and version, holding a reference to it)."""
    return _wt_cache.get("qkv", wq, lambda t: torch.cat([t.detach(), wk.detach(), wv.detach()], dim=0).contiguous())


def attention_supported(q, k, heads):
    D = q.shape[-1] // heads
    return (not _DISABLED and q.is_cuda and q.dtype == torch.float16 and D in (40, 64, 80, 160) and q.shape[1] % 32 == 0 and
            k.shape[1] >= 1 and _kv_rows(q) is not None and _kv_rows(k) is not None and
            not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad)))


def attention(q, k, v, heads, k2=None, v2=None, weight2=1.0):
    """softmax(q k^T / sqrt(D)) v [+ weight2 * softmax(q k2^T / sqrt(D)) v2] on [B, N, heads * D] projections; returns
    [B, Nq, heads * D] (csrc/attention.hip).  Key counts are arbitrary (77 text tokens, 4 image tokens, 2N mutual keys).
    k / v (and k2 / v2) may be column ranges of one wide projection matrix (same row stride for the pair)."""
    B, Nq, C = q.shape
    D = C // heads
    o = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    ld_q = _kv_rows(q)                     # q may be a column range of a fused q | k | v projection (row stride 3 C)
    if ld_q is None:
        q, ld_q = q.contiguous(), C
    null = ctypes.c_void_p(None)
    two = k2 is not None

    def pair(a, b):
        la, lb = _kv_rows(a), _kv_rows(b)
        if la is None or la != lb:
            a, b = a.contiguous(), b.contiguous()
            la = C
        return a, b, la
    k, v, ld = pair(k, v)
    ld2 = C
    if two:
        k2, v2, ld2 = pair(k2, v2)
    rc = _lib.nn_lib().gip_attention_fwd_strided2_f16(_p(q), _p(k), _p(v), _p(o), B, heads, Nq, k.shape[1], D, float(D) ** -0.5,
                                                     _p(k2) if two else null, _p(v2) if two else null,
                                                     k2.shape[1] if two else 0, float(weight2), ld_q, ld, ld2,
                                                     ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_attention_fwd_strided2_f16 failed with status %d" % rc)
    return o


class _WideHeadAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(D)) v for ONE wide head (the VAE encoder's 512-channel mid attention, 4096 tokens): the two products
    forward and four backward are dense GEMMs at the FLOP minimum (hipBLASLt through torch.bmm / baddbmm — a flash-style kernel
    would re-compute the scores in the backward and a 512-wide head does not fit a wave's accumulators); the softmax between
    them and its backward are this repo's in-place row kernels (csrc/softmax.hip): no `q * scale` pass, no second score tensor,
    no softmax kernel of the framework.  q, k, v [B, N, D] half."""

    @staticmethod
    def forward(ctx, q, k, v):
        scale = float(q.shape[-1]) ** -0.5
        fallback("VAE mid attention (dense GEMMs around the own softmax kernels)", q, library=True)
        p = torch.bmm(q, k.transpose(1, 2))                       # raw scores [B, N, N]; becomes P in place
        rc = _lib.nn_lib().gip_softmax_rows_f16(_p(p), p.shape[0] * p.shape[1], p.shape[2], scale,
                                                ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_softmax_rows_f16 failed with status %d" % rc)
        ctx.save_for_backward(q, k, v, p)
        ctx.scale = scale
        return torch.bmm(p, v)

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        do = do.contiguous()
        dv = torch.bmm(p.transpose(1, 2), do)
        dp = torch.bmm(do, v.transpose(1, 2))                     # becomes dL/d(scale * scores) in place
        # the 1 / sqrt(D) factor is NOT folded into the half-rounded score gradient (P (dP - sum dP P) is ~1e-4 dP already: times
        # 0.044 it would sink into fp16's subnormals — measured: dL/dimage 2.6e-3 -> 5.6e-3 from fp32 at loss scale 1); it rides as
        # the float32 alpha of the two products that consume it
        rc = _lib.nn_lib().gip_softmax_rows_backward_f16(_p(p), _p(dp), p.shape[0] * p.shape[1], p.shape[2], 1.0,
                                                         ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_softmax_rows_backward_f16 failed with status %d" % rc)
        dq = torch.baddbmm(q, dp, k, beta=0.0, alpha=ctx.scale)
        dk = torch.baddbmm(k, dp.transpose(1, 2), q, beta=0.0, alpha=ctx.scale)
        return dq, dk, dv


def wide_head_attention_supported(q, k):
    return (not _DISABLED and q.is_cuda and q.dtype == torch.float16 and q.dim() == 3 and q.is_contiguous() and k.is_contiguous() and
            k.shape[1] % 8 == 0 and k.shape[1] <= 8192)


def wide_head_attention(q, k, v):
    return _WideHeadAttention.apply(q, k, v.contiguous())


def linear_supported(x, w):
    return (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and w.dtype == torch.float16 and x.is_contiguous() and
            w.is_contiguous() and x.shape[-1] % 64 == 0 and x.shape[-1] >= 64 and
            not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)) and
            x.numel() * 2 < (1 << 31) and (x.numel() // x.shape[-1]) * w.shape[0] * 2 < (1 << 31))


_ROWS_ATTR = "_gip_row_stats"


def row_stats(x):
    """[M, parts, 2] float32 per-row (sum, sum of squares) partials that the MFMA linear which produced `x` left in its epilogue
    (attached to the tensor OBJECT: any view / copy drops it), or None."""
    rs = getattr(x, _ROWS_ATTR, None)
    if rs is None or rs.dim() != 3 or rs.shape[0] != x.numel() // x.shape[-1] or os.environ.get("GIP_LN_FOLD", "1") == "0":
        return None
    return rs


def _ln_fold_weights(norm, w, bias):
    """(W * gamma as half, s = row sums of that matrix, t = W beta + b) of LayerNorm `norm` folded into the projection (w, bias):
    LN(x) W^T + b = rstd (x (W gamma)^T) - rstd mu s + t.  Frozen weights: cached with the derived convolution weights."""
    tag = "lnf:%d:%d:%d:%d:%d" % (norm.weight.data_ptr(), norm.weight._version, norm.bias.data_ptr(), norm.bias._version,
                                 0 if bias is None else bias.data_ptr())

    def make(wt):
        wf = wt.detach().float()
        wg = (wf * norm.weight.detach().float()[None, :]).to(wt.dtype).contiguous()
        s_vec = wg.float().sum(dim=1).contiguous()
        t_vec = wf @ norm.bias.detach().float()
        if bias is not None:
            t_vec = t_vec + bias.detach().float()
        return wg, s_vec, t_vec.contiguous()
    return _wt_cache.get(tag, w, make)


def linear_ln(x, norm, w, bias=None, geglu_act=False):
    """F.linear(norm(x), w, bias) — or GEGLU of it — with the LayerNorm FOLDED into the GEMM (gip_linear_ln_f16): x is read raw,
    its row statistics come from the partial sums its producer attached (row_stats).  Returns None when the fold does not apply
    (no row statistics, a shape the own kernel does not take or loses to hipBLASLt): the caller then runs norm + projection."""
    rows = row_stats(x)
    n_out = w.shape[0] // 2 if geglu_act else w.shape[0]
    M, K = x.numel() // x.shape[-1], x.shape[-1]
    if (rows is None or not linear_supported(x, w) or K != norm.normalized_shape[0] or norm.weight is None or norm.bias is None or
            norm.weight.dtype != torch.float16 or n_out % (64 if geglu_act else 8) or
            not (geglu_act or linear_prefers_own(M, K, n_out))):
        return None
    wg, s_vec, t_vec = _ln_fold_weights(norm, w, bias)
    out = torch.empty(x.shape[:-1] + (n_out,), dtype=x.dtype, device=x.device)
    rc = _lib.nn_lib().gip_linear_ln_f16(_p(x), _p(wg), _p(s_vec), _p(t_vec), _p(out), M, K, n_out, int(geglu_act), _p(rows), rows.shape[1],
                                         float(norm.eps), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_linear_ln_f16 failed with status %d" % rc)
    return out


class LNInput:
    """`norm(x)` not yet computed: a projection that can fold the LayerNorm takes (x, norm) as they are (linear_ln), everything
    else calls materialize() — the LayerNorm kernel, once."""

    def __init__(self, x, norm):
        self.x, self.norm, self._y = x, norm, None

    def materialize(self):
        if self._y is None:
            self._y = self.norm(self.x)
        return self._y

    def linear(self, w, bias=None):
        y = linear_ln(self.x, self.norm, w, bias)
        return y if y is not None else linear_auto(self.materialize(), w, bias)


def linear_prefers_own(M, K, N):
    """Which dense GEMM shapes run on this repo's MFMA linear (conv3x3_kernel<TAPS = 1>) and which stay on hipBLASLt — by
    measurement, shape by shape (profiles/r04_gemm_own_vs_hipblaslt.txt = tools/exp_gemm_table.py: every GEMM of the denoise at
    batch 12 / 6 / 3, same process, alternating).  The own kernel (128-row tiles, K steps of 64, bias / residual in its
    epilogue, no split-K) wins the short-K, narrow-N layers — proj_in, to_q, q|k|v at 64^2, the out-projections, the 1x1
    shortcuts and zero convolutions: 0.48-0.86 of the library's time — and loses the weight-heavy ones (wide N at few rows:
    ff_in below 64^2, q|k|v at 16^2, K >= 2560 at <= 6144 rows: 1.1-2.2x), where the library's 256 x 256 tiles and split-K win.
    GIP_OWN_GEMM=0 sends everything that has no fused epilogue to the library, =2 everything supported to the own kernel."""
    mode = os.environ.get("GIP_OWN_GEMM", "1")
    if mode == "0":
        return False
    if mode == "2":
        return True
    return M >= 49152 or (K <= 1280 and N <= 1280) or (K <= 2560 and N <= 640 and M >= 12288)


def linear_auto(x, w, bias=None, residual=None, want_rows=False):
    """F.linear(x, w, bias) (+ residual) on whichever of the two GEMM paths is faster for the shape (linear_prefers_own).
    `want_rows`: the result feeds a LayerNorm — on the own kernel its epilogue also leaves the per-row sums that let the consumer
    GEMM fold that LayerNorm (attached to the result: row_stats)."""
    M = x.numel() // x.shape[-1]
    if linear_supported(x, w) and w.shape[0] % 4 == 0 and (residual is None or residual.is_contiguous()) and \
            linear_prefers_own(M, x.shape[-1], w.shape[0]):
        if want_rows and w.shape[0] % 8 == 0 and os.environ.get("GIP_LN_FOLD", "1") != "0":
            holder = []
            out = linear(x, w, bias, residual, rows=holder)
            if holder:
                setattr(out, _ROWS_ATTR, holder[0])
            return out
        return linear(x, w, bias, residual)
    fallback("linear_auto (measured dispatch to hipBLASLt)" if linear_supported(x, w) else "linear_auto", x, library=linear_supported(x, w))
    y = F.linear(x, w, bias)
    return y if residual is None else y + residual


def linear(x, w, bias=None, residual=None, geglu_act=False, stats=None, rows=None):
    """F.linear(x, w, bias) (+ residual) or, with geglu_act, GEGLU(F.linear(x, w, bias)) — one MFMA kernel with the bias /
    residual / activation in its epilogue (csrc/conv3x3.hip, TAPS = 1).  Inference only (frozen denoiser under no_grad);
    anything else goes to hipBLASLt through F.linear."""
    n_out = w.shape[0] // 2 if geglu_act else w.shape[0]
    if linear_supported(x, w) and n_out % (64 if geglu_act else 4) == 0 and (residual is None or residual.is_contiguous()):
        M = x.numel() // x.shape[-1]
        out = torch.empty(x.shape[:-1] + (n_out,), dtype=x.dtype, device=x.device)
        null = ctypes.c_void_p(None)
        if rows is not None and not geglu_act and stats is None and n_out % 8 == 0:
            # `rows`: a list that receives the per-row (sum, sum of squares) partials of the output [M, parts, 2]
            lib = _lib.nn_lib()
            rt = torch.empty((M, lib.gip_linear_row_parts(M, n_out), 2), dtype=torch.float32, device=x.device)
            rc = lib.gip_linear_rows_f16(_p(x), _p(w), null if bias is None else _p(bias), null if residual is None else _p(residual), _p(out),
                                         M, x.shape[-1], n_out, _p(rt), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("gip_linear_rows_f16 failed with status %d" % rc)
            rows.append(rt)
            return out
        if stats is not None and not geglu_act and M % 128 == 0 and n_out % 8 == 0 and os.environ.get("GIP_GN_STATS", "1") != "0":
            st = torch.empty((M // 128, n_out, 2), dtype=torch.float32, device=x.device)
            rc = _lib.nn_lib().gip_linear_stats_f16(_p(x), _p(w), null if bias is None else _p(bias),
                                                    null if residual is None else _p(residual), _p(out), M, x.shape[-1], n_out, _p(st),
                                                    ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("gip_linear_stats_f16 failed with status %d" % rc)
            stats.append(st)
            return out
        rc = _lib.nn_lib().gip_linear_f16(_p(x), _p(w), null if bias is None else _p(bias), null if residual is None else _p(residual),
                                          _p(out), M, x.shape[-1], n_out, int(geglu_act),
                                          ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_linear_f16 failed with status %d" % rc)
        return out
    fallback("linear", x)
    y = F.linear(x, w, bias)
    if geglu_act:
        return geglu(y)
    return y if residual is None else y + residual


# ---------------------------------------------------------------------------------------------------------------------
# ResnetBlock2D with frozen weights as ONE autograd node (the differentiable VAE encoder)
# ---------------------------------------------------------------------------------------------------------------------
def _gn_fwd_raw(x, gn, addend, chan_stats):
    """(y, mean, rstd) of GroupNormAct `gn` on NHWC fp16 x through the C-ABI (no autograd)."""
    N, C, H, W = x.shape
    lib = _lib.nn_lib()
    ad_ptr, ad_stride = ctypes.c_void_p(None), 0
    if addend is not None:
        ad_ptr, ad_stride = _p(addend), (addend.stride(0) if addend.dim() == 2 and addend.shape[0] > 1 else 0)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    mean = torch.empty((N, gn.num_groups), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    if chan_stats is not None:
        rc = lib.gip_gn_silu_forward_stats(_p(x), _p(gn.weight), _p(gn.bias), _p(y), _p(mean), _p(rstd), N, H * W, C, gn.num_groups,
                                           float(gn.eps), int(gn.act), ad_ptr, ad_stride, _p(chan_stats), chan_stats.shape[0] // N, stream)
    else:
        ws = _workspace(x.device, lib.gip_gn_workspace_bytes(N, gn.num_groups))
        rc = lib.gip_gn_silu_forward(_p(x), _p(gn.weight), _p(gn.bias), _p(y), _p(mean), _p(rstd), N, H * W, C, gn.num_groups,
                                     float(gn.eps), int(gn.act), ad_ptr, ad_stride, _p(ws), ws.numel(), stream)
    if rc != 0:
        raise RuntimeError("GroupNorm forward failed with status %d" % rc)
    return y, mean, rstd


def _conv_gn_in(x, gn, addend, chan_stats, w, bias, residual, stats):
    """(conv3x3(gn(x + addend)), mean, rstd) with the GroupNorm (+ SiLU) applied inside the convolution's halo load
    (gip_conv3x3_gnin_nhwc_f16): statistics from the producer's partial sums (one tiny launch), no apply pass, no normalised tensor.
    None when the layer does not qualify (the caller then runs the apply pass and the plain convolution).  `stats`: a list that
    receives the OUTPUT's chan_stats.  GIP_CONV_GNIN=0 switches it off (same-box A/B)."""
    N, C, H, W = x.shape
    cout = w.shape[0]
    # (Cout % 256 == 0 layers run on the 256-wide tile, which has no halo mode and is faster for them than this kernel)
    if (chan_stats is None or _DISABLED or C != 128 or H % 8 or W % 16 or cout % 8 or cout % 256 == 0 or C % gn.num_groups or
            _conv_tiles(N, H, W, cout) < 256 or os.environ.get("GIP_CONV_GNIN", "1") == "0" or os.environ.get("GIP_CONV_HALO", "1") == "0" or
            x.numel() * 2 >= (1 << 31) or N * H * W * cout * 2 >= (1 << 31)):
        return None
    lib = _lib.nn_lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    null = ctypes.c_void_p(None)
    ad_ptr, ad_stride = null, 0
    if addend is not None:
        ad_ptr, ad_stride = _p(addend), (addend.stride(0) if addend.dim() == 2 and addend.shape[0] > 1 else 0)
    mean = torch.empty((N, gn.num_groups), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    rc = lib.gip_gn_stats_from_partials(_p(mean), _p(rstd), N, H * W, C, gn.num_groups, float(gn.eps), ad_ptr, ad_stride, _p(chan_stats),
                                        chan_stats.shape[0] // N, stream)
    if rc != 0:
        raise RuntimeError("gip_gn_stats_from_partials failed with status %d" % rc)
    out = torch.empty((N, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    st = None
    if stats is not None and stats_wanted(N, H, W, cout):
        st = torch.empty((N * H * W // 128, cout, 2), dtype=torch.float32, device=x.device)
    rc = lib.gip_conv3x3_gnin_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias), null if residual is None else _p(residual), _p(out),
                                       N, H, W, C, cout, _p(gn.weight), _p(gn.bias), _p(mean), _p(rstd), gn.num_groups, int(gn.act),
                                       ad_ptr, ad_stride, null if st is None else _p(st), stream)
    if rc == 1:
        return None                      # a shape the halo kernel does not take after all
    if rc != 0:
        raise RuntimeError("gip_conv3x3_gnin_nhwc_f16 failed with status %d" % rc)
    if st is not None:
        stats.append(st)
    return out, mean, rstd


def _dgrad_with_gn_sums(dy, w, x_gn, gn, mean, rstd, addend):
    """(dL/dy_gn, chan_sums): the data gradient conv3x3(dy, w^T-flipped) of a convolution whose input was gn(x_gn), with the two
    reductions of that GroupNorm's backward taken in the kernel's epilogue (gip_conv3x3_gnbwd_nhwc_f16); chan_sums is None
    when the shape does not qualify (the caller's GroupNorm backward then takes its own reduction pass).
    GIP_GN_BWD_SUMS=0 switches it off.  It first measured neutral (the epilogue's extra read of the GroupNorm input and its
    ~20 vector instructions per element (dsilu) cost the data-gradient kernel what the separate reduction pass took); with
    the GroupNorm-input rows requested before the accumulators are staged through LDS (like the residual rows) it is worth
    0.09 ms of the VAE's forward + backward (14.155 -> 14.065 ms, four same-box runs each)."""
    wt = _transposed_weight(w)
    N, C, H, W = x_gn.shape
    if (os.environ.get("GIP_GN_BWD_SUMS", "1") == "0" or _DISABLED or (H * W) % 256 or C % 8 or _conv_tiles(N, H, W, C) < _GN_SUMS_MIN_TILES or
            C // gn.num_groups > 256):
        return _conv_call(dy, wt, C), None
    out = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
    sums = torch.empty((N * H * W // 128, C, 2), dtype=torch.float32, device=dy.device)
    ad_ptr, ad_stride = ctypes.c_void_p(None), 0
    if addend is not None:
        ad_ptr, ad_stride = _p(addend), (addend.stride(0) if addend.dim() == 2 and addend.shape[0] > 1 else 0)
    rc = _lib.nn_lib().gip_conv3x3_gnbwd_nhwc_f16(_p(dy), _p(wt), _p(out), N, H, W, dy.shape[1], C, _p(x_gn), _p(gn.weight), _p(gn.bias),
                                                  _p(mean), _p(rstd), gn.num_groups, int(gn.act), ad_ptr, ad_stride, _p(sums),
                                                  ctypes.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_conv3x3_gnbwd_nhwc_f16 failed with status %d" % rc)
    return out, sums


def _gn_bwd_raw(x, dy, gn, mean, rstd, addend, accum=None, chan_sums=None):
    """dL/dx of the same GroupNorm (+ `accum`, the other gradient reaching x, in the same pass)."""
    N, C, H, W = x.shape
    lib = _lib.nn_lib()
    ad_ptr, ad_stride = ctypes.c_void_p(None), 0
    if addend is not None:
        ad_ptr, ad_stride = _p(addend), (addend.stride(0) if addend.dim() == 2 and addend.shape[0] > 1 else 0)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    ws = _workspace(x.device, lib.gip_gn_workspace_bytes(N, gn.num_groups))
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    if chan_sums is not None:
        rc = lib.gip_gn_silu_backward_sums(_p(x), _p(dy), _p(gn.weight), _p(gn.bias), _p(mean), _p(rstd), _p(dx), N, H * W, C,
                                           gn.num_groups, int(gn.act), ad_ptr, ad_stride,
                                           ctypes.c_void_p(None) if accum is None else _p(accum), _p(chan_sums), (H * W) // 128,
                                           _p(ws), ws.numel(), stream)
    elif accum is None:
        rc = lib.gip_gn_silu_backward(_p(x), _p(dy), _p(gn.weight), _p(gn.bias), _p(mean), _p(rstd), _p(dx), N, H * W, C,
                                      gn.num_groups, int(gn.act), ad_ptr, ad_stride, _p(ws), ws.numel(), stream)
    else:
        rc = lib.gip_gn_silu_backward_accum(_p(x), _p(dy), _p(gn.weight), _p(gn.bias), _p(mean), _p(rstd), _p(dx), N, H * W, C,
                                            gn.num_groups, int(gn.act), ad_ptr, ad_stride, _p(accum), _p(ws), ws.numel(), stream)
    if rc != 0:
        raise RuntimeError("GroupNorm backward failed with status %d" % rc)
    return dx


def _dgrad_ok(x_shape, w):
    N, _, H, W = x_shape
    return w.shape[0] % 64 == 0 and w.shape[1] % 4 == 0 and _conv_tiles(N, H, W, w.shape[1]) >= _MIN_CONV_TILES


def resblock_grad_supported(x, block):
    """The whole-block autograd node applies: fp16 NHWC input that needs a gradient, frozen fp16 weights, no time embedding
    (the VAE encoder's ResnetBlock2D), both convolutions and both data gradients on the MFMA kernel."""
    c1, c2 = block.conv1.weight, block.conv2.weight
    return (os.environ.get("GIP_RESBLOCK_NODE", "1") != "0" and fusable(x) and torch.is_grad_enabled() and x.requires_grad and
            block.time_emb_proj is None and not c1.requires_grad and c1.dtype == torch.float16 and
            c1.is_contiguous(memory_format=torch.channels_last) and c2.is_contiguous(memory_format=torch.channels_last) and
            x.shape[1] % 64 == 0 and c1.shape[0] % 64 == 0 and x.numel() * 2 < (1 << 31) and
            x.shape[0] * x.shape[2] * x.shape[3] * c1.shape[0] * 2 < (1 << 31) and
            _conv_tiles(x.shape[0], x.shape[2], x.shape[3], c1.shape[0]) >= _MIN_CONV_TILES and
            _dgrad_ok(x.shape, c1) and _dgrad_ok((x.shape[0], c1.shape[0], x.shape[2], x.shape[3]), c2) and
            block.norm1.weight.dtype == torch.float16 and not block.norm1.weight.requires_grad)


class _ResBlockNode(torch.autograd.Function):
    """ResnetBlock2D (GroupNorm+SiLU -> conv1 -> GroupNorm+SiLU -> conv2, + shortcut) with frozen weights, differentiable
    with respect to x only.  One node instead of five, so that its backward can do what autograd cannot: x receives two
    gradients (through norm1 and through the shortcut) and their sum rides in the GroupNorm backward's apply pass
    (gip_gn_silu_backward_accum) instead of a separate pass over the tensor; nothing but x, conv1's output and the four
    statistics vectors is kept for the backward."""

    @staticmethod
    def forward(ctx, x, block, stats_out):
        c1, c2 = block.conv1, block.conv2
        h1_stats = []
        # GroupNorm + SiLU applied INSIDE the convolution that consumes it where the halo-resident kernel takes the layer (Cin = 128:
        # the VAE encoder's first level, 268 MB tensors): the normalised tensor is never written or read (_conv_gn_in)
        r = _conv_gn_in(x, block.norm1, None, producer_stats(x), c1.weight, None, None, h1_stats)
        if r is not None:
            h, mean1, rstd1 = r
        else:
            y1, mean1, rstd1 = _gn_fwd_raw(x, block.norm1, None, producer_stats(x))
            h = _conv_call(y1, c1.weight, c1.weight.shape[0], None, None, h1_stats)
            del y1
        if block.conv_shortcut is None:
            res = x
        else:
            res = conv1x1(x, block.conv_shortcut.weight, block.conv_shortcut.bias)
        res = res.contiguous(memory_format=torch.channels_last)
        r = _conv_gn_in(h, block.norm2, c1.bias, h1_stats[0] if h1_stats else None, c2.weight, c2.bias, res, stats_out)
        if r is not None:
            out, mean2, rstd2 = r
        else:
            y2, mean2, rstd2 = _gn_fwd_raw(h, block.norm2, c1.bias, h1_stats[0] if h1_stats else None)      # conv1's bias enters as the addend
            out = _conv_call(y2, c2.weight, c2.weight.shape[0], c2.bias, res, stats_out)
        ctx.save_for_backward(x, h, mean1, rstd1, mean2, rstd2)
        ctx.block = block
        return out

    @staticmethod
    def backward(ctx, dy):
        x, h, mean1, rstd1, mean2, rstd2 = ctx.saved_tensors
        block = ctx.block
        c1, c2 = block.conv1, block.conv2
        dy = dy.contiguous(memory_format=torch.channels_last)
        # each data-gradient convolution also takes the two reductions of the GroupNorm backward that consumes its output
        d_y2, sums2 = _dgrad_with_gn_sums(dy, c2.weight, h, block.norm2, mean2, rstd2, c1.bias)
        d_h = _gn_bwd_raw(h, d_y2, block.norm2, mean2, rstd2, c1.bias, chan_sums=sums2)
        del d_y2
        d_y1, sums1 = _dgrad_with_gn_sums(d_h, c1.weight, x, block.norm1, mean1, rstd1, None)
        del d_h
        if block.conv_shortcut is None:
            short = dy
        else:
            ws_ = block.conv_shortcut.weight
            wt = _wt_cache.get("1x1t", ws_, lambda t: t.detach().reshape(t.shape[0], t.shape[1]).t().contiguous())   # [Cin, Cout]
            N, Co, H, W = dy.shape
            fallback("resblock shortcut data gradient", dy, library=True)
            short = F.linear(dy.permute(0, 2, 3, 1).reshape(N * H * W, Co), wt).view(N, H, W, wt.shape[0]).permute(0, 3, 1, 2)
        return _gn_bwd_raw(x, d_y1, block.norm1, mean1, rstd1, None, accum=short, chan_sums=sums1), None, None


def resblock_with_grad(x, block):
    holder = []
    out = _ResBlockNode.apply(x, block, holder)
    return attach_stats(out, holder[0] if holder else None)


def _conv_s2_supported(x, w):
    return (fusable(x) and x.shape[1] % 64 == 0 and w.shape[0] % 4 == 0 and w.dtype == torch.float16 and
            w.is_contiguous(memory_format=torch.channels_last) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and
            _conv_tiles(x.shape[0], x.shape[2] // 2, x.shape[3] // 2, w.shape[0]) >= _MIN_CONV_TILES and x.numel() * 2 < (1 << 31))


def _conv_s2_call(x, w, bias, pad, stats=None):
    """3x3 / stride 2 through the MFMA kernel; pad = 1 (symmetric padding=1) or 0 (F.pad(x, (0, 1, 0, 1)) form).
    `stats`: a list that receives the output's chan_stats (see producer_stats) when the shape qualifies (whole-K tiles)."""
    N, C, H, W = x.shape
    cout = w.shape[0]
    out = torch.empty((N, cout, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    null = ctypes.c_void_p(None)
    if (stats is not None and _conv_tiles(N, H // 2, W // 2, cout) >= 256 and ((H // 2) * (W // 2)) % 128 == 0 and cout % 8 == 0 and
            cout // 32 <= 256 and os.environ.get("GIP_GN_STATS", "1") != "0" and os.environ.get("GIP_CONV_S2_STATS", "1") != "0"):
        st = torch.empty((N * (H // 2) * (W // 2) // 128, cout, 2), dtype=torch.float32, device=x.device)
        rc = _lib.nn_lib().gip_conv3x3s2_stats_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias), _p(out), N, H, W, C, cout, pad, pad,
                                                        _p(st), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_conv3x3s2_stats_nhwc_f16 failed with status %d" % rc)
        stats.append(st)
        return out
    ws = _workspace(x.device, _SPLITK_WS_BYTES) if _conv_tiles(N, H // 2, W // 2, cout) < 256 else None
    rc = _lib.nn_lib().gip_conv3x3s2_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias), _p(out), N, H, W, C, cout, pad, pad,
                                              null if ws is None else _p(ws), 0 if ws is None else ws.numel(),
                                              ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_conv3x3s2_nhwc_f16 failed with status %d" % rc)
    return out


def downsample_sym(x, w, bias):
    """diffusers Downsample2D of the U-Net / ControlNet: 3x3, stride 2, padding 1 (frozen, no gradient path needed)."""
    if _conv_s2_supported(x, w) and not (torch.is_grad_enabled() and x.requires_grad):
        holder = []
        return attach_stats(_conv_s2_call(x, w, bias, 1, holder), holder[0] if holder else None)
    fallback("downsample_sym", x)
    return F.conv2d(x, w, bias, stride=2, padding=1)


def _upsample_conv_weight(w):
    """wt4 [4][Cout][3][3][Cin] of gip_upsample2x_conv3x3_nhwc_f16 from the convolution weight w [Cout, Cin, 3, 3]: per output
    parity the taps that read the same source pixel are summed (fp32, one rounding to half)."""
    acc = torch.float64 if w.dtype == torch.float64 else torch.float32
    wk = w.detach().to(acc).permute(0, 2, 3, 1)                # [co][ky][kx][ci]
    sets = ({0: (0,), 1: (1, 2)}, {1: (0, 1), 2: (2,)})         # parity -> {tap (input offset tap - 1): the ky it gathers}
    out = torch.zeros((4, w.shape[0], 3, 3, w.shape[1]), dtype=acc, device=w.device)
    for pi in range(2):
        for pj in range(2):
            for dy, kys in sets[pi].items():
                for dx, kxs in sets[pj].items():
                    out[2 * pi + pj, :, dy, dx, :] = sum(wk[:, ky, kx, :] for ky in kys for kx in kxs)
    return out.to(w.dtype).contiguous()


def upsample2x_conv3x3(x, w, bias):
    """conv3x3(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias) — diffusers Upsample2D.  On the GPU the four
    output parity classes run as 2 x 2-tap convolutions over x itself (summed weights, see _upsample_conv_weight): 4 / 9 of
    the FLOPs and no upsampled tensor.  The summed weights are rounded to half once: results differ from the two-step form
    by fp16 weight rounding (~2^-11 relative per weight), inside the tolerance of the fp16 convolution itself."""
    if (fusable(x) and x.shape[1] % 64 == 0 and w.shape[0] % 8 == 0 and w.dtype == torch.float16 and not w.requires_grad and
            (bias is None or not bias.requires_grad) and not (torch.is_grad_enabled() and x.requires_grad) and
            4 * _conv_tiles(x.shape[0], x.shape[2], x.shape[3], w.shape[0]) >= _UPCONV_MIN_TILES and
            x.shape[0] * 4 * x.shape[2] * x.shape[3] * max(x.shape[1], w.shape[0]) * 2 < (1 << 31)):
        N, C, H, W = x.shape
        out = torch.empty((N, w.shape[0], 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        rc = _lib.nn_lib().gip_upsample2x_conv3x3_nhwc_f16(_p(x), _p(_wt_cache.get("up4", w, _upsample_conv_weight)),
                                                           ctypes.c_void_p(None) if bias is None else _p(bias), _p(out), N, H, W, C,
                                                           w.shape[0], ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_upsample2x_conv3x3_nhwc_f16 failed with status %d" % rc)
        return out
    return conv3x3(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, bias)


def _s2_dgrad_weight(w):
    """wt4 [4][Cin][3][3][Cout] of gip_conv3x3s2_dgrad_nhwc_f16 from the forward weight w [Cout, Cin, 3, 3]."""
    wt = w.detach().permute(1, 2, 3, 0)                       # [ci][ky][kx][co]
    out = torch.zeros((4, w.shape[1], 3, 3, w.shape[0]), dtype=w.dtype, device=w.device)
    for pi in range(2):
        for pj in range(2):
            for dy, ky in ((0, 2), (1, pi)) if pi == 0 else ((1, 1),):
                for dx, kx in ((0, 2), (1, pj)) if pj == 0 else ((1, 1),):
                    out[2 * pi + pj, :, dy, dx, :] = wt[:, ky, kx, :]
    return out.contiguous()


class _DownsampleAsym(torch.autograd.Function):
    """VAE Downsample2D: F.pad(x, (0, 1, 0, 1)) -> 3x3 / stride 2 / pad 0 convolution.  Forward stays on MIOpen; the
    DATA GRADIENT is the stride-1 MFMA convolution of the zero-dilated upstream gradient with the flipped-transposed
    weight: dx[i, j] = sum_{ky, kx} dy[(i - ky) / 2, (j - kx) / 2] w[:, :, ky, kx] over even (i - ky), (j - kx), which is
    a pad-1 3x3 correlation over U[2y + 1, 2x + 1] = dy[y, x] (zeros elsewhere).  4x the minimal FLOPs, but on the
    128-channel 512^2 level it replaces a 54 TFLOP/s library kernel (1.44 ms) by ~0.5 ms."""

    @staticmethod
    def forward(ctx, x, w, bias, stats_out=None):
        ctx.save_for_backward(w)
        ctx.x_shape = tuple(x.shape)
        if _conv_s2_supported(x, w):
            return _conv_s2_call(x, w, bias, 0, stats_out)
        fallback("downsample_asym", x)
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, bias, stride=2)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        N, C, H, W = ctx.x_shape
        if (w.shape[0] % 64 == 0 and C % 8 == 0 and N * H * W * max(C, w.shape[0]) * 2 < (1 << 31)):
            # four parity classes of dx, each a small convolution over dy's grid (4 / 2 / 2 / 1 taps): minimal FLOPs
            dy = dy.contiguous(memory_format=torch.channels_last)
            dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
            rc = _lib.nn_lib().gip_conv3x3s2_dgrad_nhwc_f16(_p(dy), _p(_wt_cache.get("s2t", w, _s2_dgrad_weight)), _p(dx), N, H // 2, W // 2,
                                                            w.shape[0], C, ctypes.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("gip_conv3x3s2_dgrad_nhwc_f16 failed with status %d" % rc)
            return dx, None, None, None
        if w.shape[0] > 128:          # measured: the library's backward-data kernels are as fast at 256 / 512 channels
            fallback("downsample_asym data gradient", dy, library=True)
            return torch.nn.grad.conv2d_input((N, C, H + 1, W + 1), w, dy, stride=2)[:, :, :H, :W], None, None, None
        up = torch.empty((N, w.shape[0], H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last).zero_()
        up[:, :, 1::2, 1::2] = dy
        return _conv_call(up, _transposed_weight(w), w.shape[1]), None, None, None


def downsample_asym(x, w, bias):
    """Differentiable VAE downsample.  The dilated-gradient route is used where it wins (measured: the 128-channel
    level; at 256 / 512 channels the library's backward-data kernels are as fast)."""
    if (fusable(x) and not w.requires_grad and w.shape[0] % 64 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and
            w.is_contiguous(memory_format=torch.channels_last)):
        holder = []
        if x.requires_grad and torch.is_grad_enabled():
            return attach_stats(_DownsampleAsym.apply(x, w, bias, holder), holder[0] if holder else None)
        if _conv_s2_supported(x, w):
            return attach_stats(_conv_s2_call(x, w, bias, 0, holder), holder[0] if holder else None)
    fallback("downsample_asym", x)
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, bias, stride=2)


def _c3_kernels_apply(x, w):
    """conv_in of the VAE encoder on its dedicated kernels (csrc/conv_small.hip): 3 input channels, 128 output channels, NHWC."""
    return (x.shape[1] == 3 and w.shape[0] == 128 and x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0 and
            x.is_contiguous(memory_format=torch.channels_last) and w.is_contiguous(memory_format=torch.channels_last) and
            x.numel() // 3 * 128 * 2 < (1 << 31))


class _ConvFewInputChannels(torch.autograd.Function):
    """3x3 / pad 1 convolution whose INPUT has very few channels (the VAE's conv_in: 3 -> 128) and its data gradient
    (128 -> 3 channels over the full-resolution image, the gradient that flows back into the rasterizer).  Both are one
    pass over the 128-channel tensor on the dedicated kernels of csrc/conv_small.hip (forward: the library route was a
    MIOpen kernel + a bias kernel + an NCHW -> NHWC copy, 0.25 ms; backward: the 128-wide MFMA tile with 4 of its 128
    output channels used, 0.32 ms; the library's backward-data kernel for this shape took 1.4 ms).  Shapes the kernels do
    not take fall back to those routes."""

    @staticmethod
    def forward(ctx, x, w, bias, stats_out=None):
        ctx.save_for_backward(w)
        ctx.c3 = _c3_kernels_apply(x, w)
        if ctx.c3:
            N, _, H, W = x.shape
            out = torch.empty((N, 128, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            null = ctypes.c_void_p(None)
            stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            if stats_out is not None and os.environ.get("GIP_GN_STATS", "1") != "0" and os.environ.get("GIP_CONV_S2_STATS", "1") != "0":
                st = torch.empty((N * H * W // 128, 128, 2), dtype=torch.float32, device=x.device)
                rc = _lib.nn_lib().gip_conv3x3_c3_fwd_stats_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias), _p(out), N, H, W, 128,
                                                                     _p(st), stream)
                stats_out.append(st)
            else:
                rc = _lib.nn_lib().gip_conv3x3_c3_fwd_nhwc_f16(_p(x), _p(w), null if bias is None else _p(bias), _p(out), N, H, W, 128, stream)
            if rc != 0:
                raise RuntimeError("gip_conv3x3_c3_fwd_nhwc_f16 failed with status %d" % rc)
            return out
        fallback("conv_in (few input channels)", x)
        return F.conv2d(x, w, bias, padding=1)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        if ctx.c3:
            # wt[c][3 ty + tx][co] = w[co][2 - ty][2 - tx][c]
            wt = _wt_cache.get("c3t", w, lambda t: t.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous())
            N, _, H, W = dy.shape
            dx = torch.empty((N, 3, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
            rc = _lib.nn_lib().gip_conv3x3_c3_dgrad_nhwc_f16(_p(dy), _p(wt), _p(dx), N, H, W, 128,
                                                             ctypes.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("gip_conv3x3_c3_dgrad_nhwc_f16 failed with status %d" % rc)
            return dx, None, None, None

        def make(t):
            wt4 = torch.zeros((4, t.shape[0], 3, 3), dtype=t.dtype, device=t.device)
            wt4[:t.shape[1]] = t.detach().flip(2, 3).transpose(0, 1)
            return wt4.contiguous(memory_format=torch.channels_last)
        wt = _wt_cache.get("few", w, make)
        return _conv_call(dy, wt, 4)[:, :w.shape[1]], None, None, None


# (Cin, Cout, stride) -> output rows per workgroup tile (the output height must be a multiple of it)
_FEWCH_SHAPES = {(3, 16, 1): 16, (3, 128, 1): 16, (16, 16, 1): 8, (16, 32, 2): 8, (32, 32, 1): 8, (32, 96, 2): 8, (96, 96, 1): 8, (96, 256, 2): 4,
                 (8, 320, 1): 8, (8, 512, 1): 4}


def conv3x3_fewch(x, w, bias, stride=1, act=False):
    """F.conv2d(x, w, bias, stride, padding=1) (+ F.silu) for the narrow layers of the ControlNet's conditioning stem on the
    kernels of csrc/conv_small.hip (bias and SiLU in the epilogue: one launch instead of the library's convolution, bias,
    layout-copy and activation kernels).  Frozen weights, no gradient path; other shapes take the library route."""
    cin, cout = x.shape[1], w.shape[0]
    Ho, Wo = x.shape[2] // stride, x.shape[3] // stride
    if (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and (cin, cout, stride) in _FEWCH_SHAPES and
            w.dtype == torch.float16 and tuple(w.shape[1:]) == (cin, 3, 3) and
            x.is_contiguous(memory_format=torch.channels_last) and w.is_contiguous(memory_format=torch.channels_last) and
            x.shape[2] % stride == 0 and x.shape[3] % stride == 0 and Wo % 16 == 0 and Ho % _FEWCH_SHAPES[(cin, cout, stride)] == 0 and
            not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))):
        out = torch.empty((x.shape[0], cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        rc = _lib.nn_lib().gip_conv3x3_fewch_nhwc_f16(_p(x), _p(w), ctypes.c_void_p(None) if bias is None else _p(bias), _p(out),
                                                      x.shape[0], x.shape[2], x.shape[3], cin, cout, stride, int(bool(act)),
                                                      ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("gip_conv3x3_fewch_nhwc_f16 failed with status %d" % rc)
        return out
    fallback("conv3x3_fewch", x)
    y = F.conv2d(x, w, bias, stride=stride, padding=1)
    return F.silu(y) if act else y


_PAD_ZEROS = {}


def conv3x3_latent_in(x, w, bias):
    """conv_in of the U-Net / ControlNet: F.conv2d(x, w, bias, padding=1) on the 4-channel latents (4 -> 320).  On the GPU the
    latents are padded to 8 channels (one concatenation with a cached block of zeros; the weight's padded copy is cached) and
    run on csrc/conv_small.hip's few-channel kernel — the library route was a MIOpen kernel + layout copies.  Frozen weights,
    no gradient path."""
    N, C, H, W = x.shape
    if (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and C == 4 and tuple(w.shape) == (320, 4, 3, 3) and w.dtype == torch.float16 and
            x.is_contiguous(memory_format=torch.channels_last) and H % 8 == 0 and W % 16 == 0 and
            not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))):
        key = (N, H, W, x.device)
        z = _PAD_ZEROS.get(key)
        if z is None:
            if len(_PAD_ZEROS) > 16:
                _PAD_ZEROS.clear()
            z = _PAD_ZEROS[key] = torch.zeros((N, 4, H, W), dtype=x.dtype, device=x.device).contiguous(memory_format=torch.channels_last)
        x8 = torch.cat([x, z], dim=1)
        if not x8.is_contiguous(memory_format=torch.channels_last):
            x8 = x8.contiguous(memory_format=torch.channels_last)

        def pad8(t):
            p = torch.zeros((t.shape[0], 8, 3, 3), dtype=t.dtype, device=t.device)
            p[:, :4] = t.detach()
            return p.contiguous(memory_format=torch.channels_last)
        return conv3x3_fewch(x8, _wt_cache.get("pad8", w, pad8), bias)
    fallback("conv_in (4 latent channels)", x)
    return F.conv2d(x, w, bias, padding=1)


def folded_quant_conv(conv_out, quant_conv):
    """(W', b') of quant_conv(conv_out(x)) as ONE 3x3 convolution: the 1x1 quant_conv (8 -> 8) of the VAE encoder composed into its
    conv_out (512 -> 8) — W'[o] = sum_m Wq[o, m] Wc[m], b' = Wq bc + bq (float32 arithmetic, one rounding to half).  Frozen weights:
    cached with the derived convolution weights, keyed by all four tensors."""
    wc, bc, wq, bq = conv_out.weight, conv_out.bias, quant_conv.weight, quant_conv.bias
    tag = "quantfold:%d:%d:%d:%d:%d:%d" % (wq.data_ptr(), wq._version, bq.data_ptr(), bq._version, bc.data_ptr(), bc._version)

    def make(t):
        q = wq.detach().float().reshape(wq.shape[0], wq.shape[1])
        w2 = torch.einsum("om,mchw->ochw", q, t.detach().float()).to(t.dtype).contiguous(memory_format=torch.channels_last)
        b2 = (q @ bc.detach().float() + bq.detach().float()).to(t.dtype).contiguous()
        return w2, b2
    return _wt_cache.get(tag, wc, make)


_NARROW_OUT_SHAPES = {(320, 4): 8, (512, 8): 4}       # (Cin, Cout) -> output rows per workgroup tile


def _narrow_out_applies(x, w):
    cin, cout = x.shape[1], w.shape[0]
    return (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and (cin, cout) in _NARROW_OUT_SHAPES and
            w.dtype == torch.float16 and tuple(w.shape[1:]) == (cin, 3, 3) and not w.requires_grad and
            x.is_contiguous(memory_format=torch.channels_last) and x.shape[3] % 16 == 0 and
            x.shape[2] % _NARROW_OUT_SHAPES[(cin, cout)] == 0)


def _narrow_out_call(x, w, bias):
    N, cin, H, W = x.shape
    cout = w.shape[0]

    def pad16(t):            # the kernel's weight block: 16 rows [co][ky][kx][ci], rows >= Cout zero
        p = torch.zeros((16, cin, 3, 3), dtype=t.dtype, device=t.device)
        p[:cout] = t.detach()
        return p.contiguous(memory_format=torch.channels_last)
    out = torch.empty((N, cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = _lib.nn_lib().gip_conv3x3_fewch_nhwc_f16(_p(x), _p(_wt_cache.get("pad16", w, pad16)), ctypes.c_void_p(None) if bias is None else _p(bias),
                                                  _p(out), N, H, W, cin, cout, 1, 0, ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_conv3x3_fewch_nhwc_f16 failed with status %d" % rc)
    return out


class _NarrowOutConv(torch.autograd.Function):
    """The differentiable form (VAE encoder conv_out): forward on the kernel, data gradient (8 -> 512 channels) on the few-channel
    kernel with the flipped-transposed weight (round 5; the library's backward-data kernel before)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(w)
        ctx.x_shape = tuple(x.shape)
        return _narrow_out_call(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        N, C, H, W = ctx.x_shape
        if (tuple(w.shape[:2]), 1) == ((8, 512), 1) and H % _FEWCH_SHAPES[(8, 512, 1)] == 0 and W % 16 == 0 and dy.dtype == torch.float16:
            return conv3x3_fewch(dy, _transposed_weight(w), None), None, None
        fallback("conv_out data gradient", dy, library=True)
        return torch.nn.grad.conv2d_input(ctx.x_shape, w, dy, padding=1), None, None


def conv3x3_narrow_out(x, w, bias):
    """F.conv2d(x, w, bias, padding=1) for the two output convolutions with very few OUTPUT channels — conv_out of the U-Net
    (320 -> 4) and of the VAE encoder (512 -> 8) — on csrc/conv_small.hip's halo-in-LDS kernel (the library needed a
    convolution, a bias and a layout-copy kernel: 0.08 / 0.05 ms)."""
    if _narrow_out_applies(x, w) and (bias is None or not bias.requires_grad):
        if torch.is_grad_enabled() and x.requires_grad:
            return _NarrowOutConv.apply(x, w, bias)
        return _narrow_out_call(x, w, bias)
    fallback("conv3x3_narrow_out", x)
    return F.conv2d(x, w, bias, padding=1)


def conv3x3_few_inputs(x, w, bias):
    if (not _DISABLED and x.is_cuda and x.dtype == torch.float16 and x.requires_grad and torch.is_grad_enabled() and
            not w.requires_grad and w.shape[1] <= 4 and w.shape[0] % 64 == 0 and
            _conv_tiles(x.shape[0], x.shape[2], x.shape[3], 4) >= _MIN_CONV_TILES and
            x.shape[0] * x.shape[2] * x.shape[3] * w.shape[0] * 2 < (1 << 31)):
        holder = []
        return attach_stats(_ConvFewInputChannels.apply(x, w, bias, holder), holder[0] if holder else None)
    fallback("conv3x3_few_inputs", x)
    return F.conv2d(x, w, bias, padding=1)
