"""Host-side mirror of the `diff_gaussian_rasterization` package surface, on top of the C-ABI HIP library.

Reference interface reproduced (the package is an un-vendored dependency; the contract is its call sites):
  * GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
    projmatrix, sh_degree, campos, prefiltered, debug)          gaussian_renderer/__init__.py:36-49
  * GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
    cov3D_precomp) -> (color [3,H,W], radii [P] int32, depth [1,H,W], alpha [1,H,W])      :51, :85-93
  * GaussianRasterizer.markVisible(positions)
Error behaviour: "Please provide exactly one of either SHs or precomputed colors!" /
"Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!" (Exception), as the fork.

Beyond the reference: `rasterize_views` renders V views of the same Gaussians in ONE launch set (the reference
loops over its 4 cameras, threestudio/systems/GaussianIP.py:154-173); V = 1 is exactly the reference call.

No host synchronisation on the training path: the reference's blocking read of `num_rendered` is replaced by a
capacity hint + a device-side overflow flag that is checked when backward starts.
An overflowed training forward never aborts the run and never desynchronises a multi-GPU job: the backward kernels
write ZERO gradients (the step degenerates to a zero-gradient step on every rank alike), the host reads the header one
step late, raises the hint, counts the event in `overflow_events` and warns.  `GIP_RASTER_ON_OVERFLOW=raise` restores
the exception.  No-grad calls (output renders) are checked immediately and re-run at the exact requirement.
"""
import ctypes
import os
import warnings
from typing import NamedTuple, Optional, Sequence

import torch

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---------------------------------------------------------------------------------------------
# capacity policy (replaces the reference's per-call D2H of num_rendered)
# ---------------------------------------------------------------------------------------------
_MIN_CAPACITY = 1 << 20
_CAPACITY_MARGIN = 65536
_capacity_hint = {}     # (device index, P, V, H, W) -> last observed num_rendered
_pending_checks = []    # [event, pinned header, key, capacity] of deferred-check forwards whose header is not read yet
overflow_events = 0     # forwards that overflowed their capacity hint (their steps ran with zero gradients)


def _on_overflow():
    return os.environ.get("GIP_RASTER_ON_OVERFLOW", "zero")


def _exact_lists():
    """GIP_RASTER_EXACT_LISTS=1: every tile of the fork's 3-sigma rectangle gets its instance (tile / index buffers
    bit-for-bit the fork's); default: only the tiles the alpha >= 1/255 region reaches (same outputs, 25 % fewer instances)."""
    return os.environ.get("GIP_RASTER_EXACT_LISTS", "0") == "1"


def _sh_scalar():
    """GIP_RASTER_SH_SCALAR=1: the SH colour contraction always runs as eval_sh's scalar chain (bit-exact against the oracle's colours).
    Default: launch sets of >= 2 views at sh_degree >= 1 contract on the matrix cores (csrc/sh_mfma.hip); colours agree to a few ulp."""
    return os.environ.get("GIP_RASTER_SH_SCALAR", "0") == "1"


def _strict():
    return os.environ.get("GIP_RASTER_SYNC", "0") == "1"


def _hint_key(dev, P, V, H, W):
    # keyed by P too: a densify / prune step changes P, and the first call with the new P sizes itself synchronously
    return (dev.index if dev.index is not None else torch.cuda.current_device(), P, V, H, W)


def _pick_capacity(key, P, V):
    last = _capacity_hint.get(key)
    if last is None:
        return None  # unknown: first call learns it synchronously
    return int(max(_MIN_CAPACITY, 3 * last + _CAPACITY_MARGIN))


_dummy = {}


def _ptr(t):
    if t is None:
        return None
    if t.numel() == 0:
        # empty tensors have a null data_ptr; the C-ABI distinguishes "absent" (null) from "empty" (P = 0)
        d = _dummy.get(t.device)
        if d is None:
            d = _dummy[t.device] = torch.zeros(64, dtype=torch.float32, device=t.device)
        return ctypes.c_void_p(d.data_ptr())
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t, name, shape_last=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("%s must be a CUDA/HIP tensor (the rasterizer has no CPU path)" % name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


class _Plan:
    """Everything one forward/backward pair shares."""
    pass


def _make_config(P, V, H, W, sh_degree, M, scale_modifier, tanfovx, tanfovy, capacity, prefiltered, debug, forward_only=False):
    cfg = _lib.GipRasterConfig()
    cfg.P, cfg.V, cfg.H, cfg.W = int(P), int(V), int(H), int(W)
    cfg.sh_degree, cfg.sh_coeffs = int(sh_degree), int(M)
    cfg.prefiltered, cfg.debug = int(bool(prefiltered)), int(bool(debug))
    cfg.scale_modifier = float(scale_modifier)
    for v in range(V):
        cfg.tanfovx[v] = float(tanfovx[v])
        cfg.tanfovy[v] = float(tanfovy[v])
    cfg.capacity = int(capacity)
    cfg.exact_lists = 1 if _exact_lists() else 0
    cfg.forward_only = 1 if forward_only else 0
    cfg.sh_scalar = 1 if _sh_scalar() else 0
    return cfg


def _check(rc, what):
    if rc != _lib.GIP_OK:
        msg = "%s failed: %s (status %d)" % (what, _lib.status_string(rc), rc)
        if rc == 1:
            raise ValueError(msg)
        raise RuntimeError(msg)


def _run_forward(plan, capacity, forward_only=False):
    lib = _lib.raster_lib()
    dev = plan.means3D.device
    cfg = _make_config(plan.P, plan.V, plan.H, plan.W, plan.sh_degree, plan.M, plan.scale_modifier, plan.tanfovx,
                       plan.tanfovy, capacity, plan.prefiltered, plan.debug, forward_only)
    nbytes = lib.gip_raster_state_bytes(ctypes.byref(cfg))
    if nbytes == 0:
        raise ValueError("invalid rasterizer configuration (P=%d V=%d H=%d W=%d sh_degree=%d)" %
                         (plan.P, plan.V, plan.H, plan.W, plan.sh_degree))
    state = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    V, P, H, W = plan.V, plan.P, plan.H, plan.W
    color = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
    depth = torch.empty((V, 1, H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((V, 1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((V, P), dtype=torch.int32, device=dev)
    ins = _lib.GipRasterInputs(_ptr(plan.means3D), _ptr(plan.shs), _ptr(plan.colors_precomp), _ptr(plan.opacities),
                               _ptr(plan.scales), _ptr(plan.rotations), _ptr(plan.cov3D_precomp),
                               _ptr(plan.viewmatrix), _ptr(plan.projmatrix), _ptr(plan.campos), _ptr(plan.bg))
    plan.header_slot, plan.host_header = _header_slot()
    outs = _lib.GipRasterOutputs(_ptr(color), _ptr(radii), _ptr(depth), _ptr(alpha),
                                 ctypes.c_void_p(plan.host_header.data_ptr()))
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.gip_raster_forward(ctypes.byref(cfg), ctypes.byref(ins), ctypes.byref(outs), _ptr(state), nbytes, stream)
    _check(rc, "gip_raster_forward")
    plan.cfg, plan.state, plan.capacity = cfg, state, capacity
    return color, radii, depth, alpha


# Pinned, device-mapped host slots the binning kernel mirrors the header into (GipRasterOutputs::host_header): the
# capacity verdict reaches the host with the event below and without a device-to-host copy in the stream (the blit
# kernel of a 64-byte copy costs ~4 us of a 0.6 ms step).  A ring, so that a slot is not rewritten while a deferred
# check of an earlier forward still has to read it.
_HEADER_SLOTS = 64
_header_ring = None
_header_next = 0
_header_owner = [None] * _HEADER_SLOTS      # pending entry that still has to read slot i


def _header_slot():
    global _header_ring, _header_next
    if _header_ring is None:
        _header_ring = torch.zeros((_HEADER_SLOTS, 16), dtype=torch.int32).pin_memory()
    i = _header_next
    _header_next = (i + 1) % _HEADER_SLOTS
    owner = _header_owner[i]
    if owner is not None and any(e is owner for e in _pending_checks):
        owner[0].synchronize()              # 64 forwards ago and still unread: settle it before the slot is reused
        _pending_checks[:] = [e for e in _pending_checks if e is not owner]
        _settle(owner)
    _header_owner[i] = None
    return i, _header_ring[i]


def _read_header_async(plan):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(plan.means3D.device))
    return ev, plan.host_header


def _settle(entry):
    """Reads a completed header copy: teaches the capacity policy the real size; reports an overflow.  Returns the
    overflow flag."""
    global overflow_events
    ev, host, key, cap = entry
    num_rendered, overflow = int(host[1]), int(host[2])
    _capacity_hint[key] = max(num_rendered, 1)
    if overflow:
        overflow_events += 1
        msg = ("gaussianip_amd rasterizer: %d tile instances exceeded the capacity hint %d; that forward's images are "
               "truncated and its backward produced zero gradients (zero-gradient step). The hint has been raised." %
               (num_rendered, cap))
        if _on_overflow() == "raise":
            raise RuntimeError(msg + " (GIP_RASTER_ON_OVERFLOW=raise)")
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return num_rendered, bool(overflow)


def _drain_pending(block=False):
    """Settles the deferred checks of earlier forwards whose header copy has landed — including grad-enabled renders
    that were never back-propagated (their check would otherwise never run)."""
    i = 0
    while i < len(_pending_checks):
        entry = _pending_checks[i]
        if block or entry[0].query():
            if block:
                entry[0].synchronize()
            _pending_checks.pop(i)
            _settle(entry)
        else:
            i += 1


def settle_pending():
    """Blocks until every deferred capacity check has been read and reported.  The check of a training forward runs when
    its backward has been enqueued, and the check of a grad-enabled forward that is never back-propagated runs at the
    next rasterizer call — so the LAST forward of a run is only reported by calling this (end of training, before a
    checkpoint).  An overflowed training step is not a no-op: its images were truncated, its rasterizer gradients are
    zero, Adam still moves by its momentum and `add_densification_stats` counts the step with zero gradients; it is
    rare by construction (the capacity is 3x the previous call's requirement) and `overflow_events` counts it."""
    _drain_pending(block=True)


def _forward_with_policy(plan, need_backward, forward_only=False):
    """`forward_only`: the state will never be handed to the backward (a render under torch.no_grad()): the kernel skips what
    only the backward reads (GipRasterConfig::forward_only).  Not set by forward_with_state, whose callers inspect the state."""
    _drain_pending()
    key = _hint_key(plan.means3D.device, plan.P, plan.V, plan.H, plan.W)
    cap = _pick_capacity(key, plan.P, plan.V)
    sync_now = _strict() or cap is None or not need_backward
    if cap is None:
        cap = max(_MIN_CAPACITY, 8 * plan.P * plan.V)
    while True:
        outs = _run_forward(plan, cap, forward_only and os.environ.get("GIP_RASTER_FORWARD_ONLY", "1") != "0")
        ev, host = _read_header_async(plan)
        if not sync_now:
            plan.pending = [ev, host, key, cap]
            _pending_checks.append(plan.pending)
            _header_owner[plan.header_slot] = plan.pending
            return outs
        ev.synchronize()
        num_rendered, overflow = int(host[1]), int(host[2])
        _capacity_hint[key] = max(num_rendered, 1)
        if not overflow:
            plan.pending = None
            plan.num_rendered = num_rendered
            return outs
        cap = int(num_rendered * 1.25) + 65536  # exact requirement is known: re-run once


def _verify_pending(plan):
    """Deferred overflow check, run once the backward kernels are enqueued (see _RasterizeGaussians.backward)."""
    entry = getattr(plan, "pending", None)
    if entry is None:
        return
    plan.pending = None
    for i, e in enumerate(_pending_checks):
        if e is entry:
            _pending_checks.pop(i)
            break
    else:
        return                      # already settled by _drain_pending
    entry[0].synchronize()
    plan.num_rendered, plan.overflowed = _settle(entry)


def _run_backward(plan, outs, g_color, g_depth, g_alpha):
    lib = _lib.raster_lib()
    dev = plan.means3D.device
    V, P, M = plan.V, plan.P, plan.M
    cfg = plan.cfg
    sbytes = lib.gip_raster_scratch_bytes(ctypes.byref(cfg))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)

    def mk(flag, shape):
        return torch.empty(shape, dtype=torch.float32, device=dev) if flag else None

    g = dict(means3D=mk(True, (P, 3)), means2D=mk(True, (V, P, 3)),
             shs=mk(plan.shs is not None, (P, max(M, 1), 3)),
             colors_precomp=mk(plan.colors_precomp is not None, (P, 3)), opacities=mk(True, (P, 1)),
             scales=mk(plan.scales is not None, (P, 3)), rotations=mk(plan.rotations is not None, (P, 4)),
             cov3D_precomp=mk(plan.cov3D_precomp is not None, (P, 6)))
    ins = _lib.GipRasterInputs(_ptr(plan.means3D), _ptr(plan.shs), _ptr(plan.colors_precomp), _ptr(plan.opacities),
                               _ptr(plan.scales), _ptr(plan.rotations), _ptr(plan.cov3D_precomp),
                               _ptr(plan.viewmatrix), _ptr(plan.projmatrix), _ptr(plan.campos), _ptr(plan.bg))
    color, depth, alpha = outs
    gin = _lib.GipRasterGradsIn(_ptr(g_color), _ptr(g_depth), _ptr(g_alpha), _ptr(alpha), _ptr(color), _ptr(depth))
    gout = _lib.GipRasterGradsOut(_ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), _ptr(g["colors_precomp"]),
                                  _ptr(g["opacities"]), _ptr(g["scales"]), _ptr(g["rotations"]),
                                  _ptr(g["cov3D_precomp"]))
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.gip_raster_backward(ctypes.byref(cfg), ctypes.byref(ins), ctypes.byref(gin), _ptr(plan.state),
                                 plan.state.numel(), _ptr(scratch), sbytes, ctypes.byref(gout), stream)
    _check(rc, "gip_raster_backward")
    return g


def _build_plan(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings_list):
    s0 = settings_list[0]
    V = len(settings_list)
    if V > _lib.GIP_MAX_VIEWS:
        raise ValueError("at most %d views per call" % _lib.GIP_MAX_VIEWS)
    plan = _Plan()
    plan.means3D = _f32c(means3D, "means3D")
    plan.shs = _f32c(shs, "shs")
    plan.colors_precomp = _f32c(colors_precomp, "colors_precomp")
    plan.opacities = _f32c(opacities, "opacities")
    plan.scales = _f32c(scales, "scales")
    plan.rotations = _f32c(rotations, "rotations")
    plan.cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
    dev = plan.means3D.device
    plan.P, plan.V = int(plan.means3D.shape[0]), V
    plan.H, plan.W = int(s0.image_height), int(s0.image_width)
    plan.sh_degree = int(s0.sh_degree)
    plan.M = 0 if plan.shs is None else int(plan.shs.shape[1])
    plan.scale_modifier = float(s0.scale_modifier)
    plan.prefiltered, plan.debug = bool(s0.prefiltered), bool(s0.debug)
    for s in settings_list[1:]:
        if (int(s.image_height), int(s.image_width), int(s.sh_degree), float(s.scale_modifier)) != \
                (plan.H, plan.W, plan.sh_degree, plan.scale_modifier):
            raise ValueError("all views of one call must share image size, sh_degree and scale_modifier")
    plan.tanfovx = [float(s.tanfovx) for s in settings_list]
    plan.tanfovy = [float(s.tanfovy) for s in settings_list]
    if V == 1:
        # same treatment as the multi-view branch: host-side camera tensors (Camera's documented no-sync path) are moved
        mv = lambda t: t.float().reshape(1, -1).to(dev, non_blocking=True).contiguous()  # noqa: E731
        plan.viewmatrix, plan.projmatrix, plan.campos = mv(s0.viewmatrix), mv(s0.projmatrix), mv(s0.campos)
    else:
        plan.viewmatrix, plan.projmatrix, plan.campos = _pack_cameras(settings_list, dev)
    plan.bg = s0.bg.float().to(dev, non_blocking=True).contiguous()
    return plan


_camera_cache = {}      # device -> (source tensors (strong references: their ids cannot be recycled), versions, packed)


def _pack_cameras(settings_list, dev):
    """[V,16] view matrices, [V,16] projection matrices and [V,3] camera centres of one launch set as three views of ONE
    packed device tensor: a single concatenation (one kernel for device-side cameras, one upload for host-side ones)
    instead of three.  The same camera tensors as in the previous call (an orbit rendered repeatedly, a fixed
    evaluation set) are recognised by identity + version counter and cost nothing."""
    V = len(settings_list)
    src = [s.viewmatrix for s in settings_list] + [s.projmatrix for s in settings_list] + [s.campos for s in settings_list]
    versions = [t._version for t in src]
    hit = _camera_cache.get(dev)
    if hit is not None and len(hit[0]) == len(src) and all(a is b for a, b in zip(hit[0], src)) and hit[1] == versions:
        packed = hit[2]
    else:
        packed = torch.cat([t.float().reshape(-1) for t in src]).to(dev, non_blocking=True)
        if packed.numel() != 35 * V:
            raise ValueError("viewmatrix / projmatrix must have 16 elements and campos 3, per view")
        _camera_cache[dev] = (src, versions, packed)
    return packed[:16 * V].view(V, 16), packed[16 * V:32 * V].view(V, 16), packed[32 * V:].view(V, 3)


class _RasterizeGaussians(torch.autograd.Function):
    """Differentiable V-view rasterization.  Inputs that are None are passed as None (the fork passes empty tensors)."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings_list,
                need_bwd=True):
        # `need_bwd` is decided by the caller (grad mode is always off inside Function.forward and
        # ctx.needs_input_grad ignores torch.no_grad()): a backward will follow, so the overflow check may be deferred
        plan = _build_plan(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings_list)
        color, radii, depth, alpha = _forward_with_policy(plan, need_bwd, forward_only=not need_bwd)
        ctx.plan = plan
        ctx.means2D_shape = None if means2D is None else tuple(means2D.shape)
        ctx.save_for_backward(color, depth, alpha)
        ctx.mark_non_differentiable(radii)
        # outputs nobody differentiated (alpha in the training step) reach backward as None = a NULL pointer in GipRasterGradsIn,
        # which the kernels treat as zero — instead of autograd filling a [V, 1, H, W] tensor of zeros for the kernel to read
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        plan = ctx.plan
        color, depth, alpha = ctx.saved_tensors

        def prep(g):
            if g is None:
                return None
            return g.float().contiguous()

        if g_color is None and g_depth is None and g_alpha is None:          # nothing upstream: the zero gradient, no launch
            z = lambda t: None if t is None else torch.zeros_like(t)  # noqa: E731
            g2d = None if ctx.means2D_shape is None else torch.zeros(ctx.means2D_shape, dtype=torch.float32, device=plan.means3D.device)
            return (z(plan.means3D), g2d, z(plan.shs), z(plan.colors_precomp), z(plan.opacities), z(plan.scales), z(plan.rotations),
                    z(plan.cov3D_precomp), None, None)
        g = _run_backward(plan, (color, depth, alpha), prep(g_color), prep(g_depth), prep(g_alpha))
        # The deferred overflow check runs AFTER the backward kernels are enqueued: waiting for the forward's header copy
        # then overlaps with GPU work instead of idling the device when backward follows forward immediately.  The
        # kernels are capacity-clamped, so running them on an overflowed forward is safe: they write zero gradients.
        _verify_pending(plan)
        g2d = None if ctx.means2D_shape is None else g["means2D"].reshape(ctx.means2D_shape)
        return (g["means3D"], g2d, g["shs"], g["colors_precomp"], g["opacities"], g["scales"], g["rotations"],
                g["cov3D_precomp"], None, None)


def _will_backward(tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _validate(shs, colors_precomp, scales, rotations, cov3D_precomp):
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """Single view; returns the reference 4-tuple with the reference shapes."""
    ins = (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    color, radii, depth, alpha = _RasterizeGaussians.apply(*ins, [raster_settings], _will_backward(ins))
    return color[0], radii[0], depth[0], alpha[0]


def rasterize_views(means3D, means2D, opacities, settings_list: Sequence[GaussianRasterizationSettings], shs=None,
                    colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
    """V views in one launch set.  means2D: [V,P,3] zero grad carrier (or None).
    Returns color [V,3,H,W], radii [V,P], depth [V,1,H,W], alpha [V,1,H,W]."""
    _validate(shs, colors_precomp, scales, rotations, cov3D_precomp)
    ins = (means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
    return _RasterizeGaussians.apply(*ins, list(settings_list), _will_backward(ins))


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            pos = _f32c(positions, "positions")
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=pos.device)
            vm = rs.viewmatrix.float().to(pos.device, non_blocking=True).contiguous()
            pm = rs.projmatrix.float().to(pos.device, non_blocking=True).contiguous()
            rc = _lib.raster_lib().gip_raster_mark_visible(
                int(pos.shape[0]), _ptr(pos), _ptr(vm), _ptr(pm), _ptr(out),
                ctypes.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream))
            _check(rc, "gip_raster_mark_visible")
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        _validate(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


# ---------------------------------------------------------------------------------------------
# state inspection (tests / parity of the tile & index buffers)
# ---------------------------------------------------------------------------------------------
def state_views(plan):
    """Decode a forward's state buffer into named torch views (no copies)."""
    lib = _lib.raster_lib()
    L = _lib.GipRasterStateLayout()
    _check(lib.gip_raster_state_layout(ctypes.byref(plan.cfg), ctypes.byref(L)), "gip_raster_state_layout")
    st = plan.state
    V, P, H, W = plan.V, plan.P, plan.H, plan.W
    T = L.tiles_x * L.tiles_y

    def view(off, count, dtype, shape):
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return st[off:off + nbytes].view(dtype).reshape(shape)

    return dict(
        header=view(L.header, 16, torch.int32, (16,)),
        records=view(L.records, V * P * 16, torch.float32, (V, P, 16)),
        records_u32=view(L.records, V * P * 16, torch.int32, (V, P, 16)),
        inst_offset=view(L.inst_offset, V * P, torch.int32, (V, P)),
        tile_count=view(L.tile_count, V * T, torch.int32, (V, T)),
        tile_start=view(L.tile_start, V * T + 1, torch.int32, (V * T + 1,)),
        keys=view(L.keys, int(plan.capacity), torch.int64, (int(plan.capacity),)),
        n_contrib=view(L.n_contrib, V * H * W, torch.int32, (V, H, W)),
        tiles_x=L.tiles_x, tiles_y=L.tiles_y)


def forward_with_state(means3D, opacities, settings_list, shs=None, colors_precomp=None, scales=None, rotations=None,
                       cov3D_precomp=None):
    """No-grad forward that also returns the plan (for state_views); used by the parity tests."""
    _validate(shs, colors_precomp, scales, rotations, cov3D_precomp)
    plan = _build_plan(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, list(settings_list))
    outs = _forward_with_policy(plan, False)
    return outs, plan


# ---------------------------------------------------------------------------------------------
# measurement helper (bench.py): live per-stage durations via the *_profiled C-ABI entry points
# ---------------------------------------------------------------------------------------------
STAGE_NAMES = ["clear", "preprocess", "scan", "scatter", "tile_sort", "render_fwd", "render_bwd", "gather_bwd"]


def profile_stages(means3D, opacities, settings_list, g_color, g_depth=None, g_alpha=None, shs=None,
                   colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, iters=10):
    """Runs forward+backward `iters` times with hipEvent pairs around every kernel stage (on the stream the kernels
    are launched on) and returns ({stage: mean ms}, num_rendered).  Synchronises; not for the training path."""
    lib = _lib.raster_lib()
    _validate(shs, colors_precomp, scales, rotations, cov3D_precomp)
    plan = _build_plan(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, list(settings_list))
    (color, radii, depth, alpha) = _forward_with_policy(plan, False)   # sizes the capacity synchronously
    dev = plan.means3D.device
    cfg = plan.cfg
    V, P, M = plan.V, plan.P, plan.M
    sbytes = lib.gip_raster_scratch_bytes(ctypes.byref(cfg))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    g = dict(means3D=f(P, 3), means2D=f(V, P, 3), shs=f(P, max(M, 1), 3) if plan.shs is not None else None,
             colors=f(P, 3) if plan.colors_precomp is not None else None, opac=f(P, 1),
             scales=f(P, 3) if plan.scales is not None else None, rots=f(P, 4) if plan.rotations is not None else None,
             cov=f(P, 6) if plan.cov3D_precomp is not None else None)
    ins = _lib.GipRasterInputs(_ptr(plan.means3D), _ptr(plan.shs), _ptr(plan.colors_precomp), _ptr(plan.opacities),
                               _ptr(plan.scales), _ptr(plan.rotations), _ptr(plan.cov3D_precomp),
                               _ptr(plan.viewmatrix), _ptr(plan.projmatrix), _ptr(plan.campos), _ptr(plan.bg))
    outs = _lib.GipRasterOutputs(_ptr(color), _ptr(radii), _ptr(depth), _ptr(alpha))
    gin = _lib.GipRasterGradsIn(_ptr(g_color), _ptr(g_depth), _ptr(g_alpha), _ptr(alpha), _ptr(color), _ptr(depth))
    gout = _lib.GipRasterGradsOut(_ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), _ptr(g["colors"]),
                                  _ptr(g["opac"]), _ptr(g["scales"]), _ptr(g["rots"]), _ptr(g["cov"]))
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    times = (ctypes.c_float * len(STAGE_NAMES))()
    acc = [0.0] * len(STAGE_NAMES)
    for it in range(iters + 1):
        for i in range(len(STAGE_NAMES)):
            times[i] = 0.0
        _check(lib.gip_raster_forward_profiled(ctypes.byref(cfg), ctypes.byref(ins), ctypes.byref(outs),
                                               _ptr(plan.state), plan.state.numel(), stream, times), "forward_profiled")
        _check(lib.gip_raster_backward_profiled(ctypes.byref(cfg), ctypes.byref(ins), ctypes.byref(gin),
                                                _ptr(plan.state), plan.state.numel(), _ptr(scratch), sbytes,
                                                ctypes.byref(gout), stream, times), "backward_profiled")
        if it > 0:  # first iteration is warm-up
            for i in range(len(STAGE_NAMES)):
                acc[i] += float(times[i])
    return {n: acc[i] / iters for i, n in enumerate(STAGE_NAMES)}, plan.num_rendered
