"""ctypes wrapper around the CPU oracle (oracle/raster_oracle.c, oracle/knn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (gaussianip_amd) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the oracle shared objects with gcc (seconds)."""
    targets = ["libraster_oracle.so", "libknn_oracle.so"]
    if force or not all(os.path.exists(os.path.join(_DIR, t)) for t in targets):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))


_lib = None
_knn = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(os.path.join(_DIR, "libraster_oracle.so"))
        _lib.oracle_create.restype = ctypes.c_void_p
        _lib.oracle_destroy.argtypes = [ctypes.c_void_p]
        _lib.oracle_num_rendered.restype = ctypes.c_uint64
        _lib.oracle_num_rendered.argtypes = [ctypes.c_void_p]
        _lib.oracle_raster_forward.restype = ctypes.c_int
        _lib.oracle_raster_backward.restype = ctypes.c_int
        _lib.oracle_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def set_threads(n):
    _load().oracle_set_threads(ctypes.c_int(int(n)))


def max_threads():
    return int(_load().oracle_max_threads())


class RasterOracle:
    """One forward (+ optional backward) of the rasterizer on CPU.  Arguments mirror
    GaussianRasterizationSettings / GaussianRasterizer.forward (gaussian_renderer/__init__.py:36-51,85-93)."""

    def __init__(self):
        self.lib = _load()
        self.ctx = ctypes.c_void_p(self.lib.oracle_create())

    def __del__(self):
        try:
            self.lib.oracle_destroy(self.ctx)
        except Exception:
            pass

    def forward(self, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                sh_degree, campos, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        self.args = dict(means3D=_f32(means3D), shs=_f32(shs), colors_precomp=_f32(colors_precomp),
                         opacities=_f32(opacities), scales=_f32(scales), rotations=_f32(rotations),
                         cov3D_precomp=_f32(cov3D_precomp), viewmatrix=_f32(viewmatrix), projmatrix=_f32(projmatrix),
                         campos=_f32(campos), bg=_f32(bg))
        a = self.args
        P = a["means3D"].shape[0]
        H, W = int(image_height), int(image_width)
        M = 0 if a["shs"] is None else a["shs"].shape[1]
        self.P, self.H, self.W, self.M, self.D = P, H, W, M, int(sh_degree)
        self.tanfovx, self.tanfovy, self.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
        color = np.zeros((3, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        depth = np.zeros((1, H, W), np.float32)
        alpha = np.zeros((1, H, W), np.float32)
        rc = self.lib.oracle_raster_forward(
            self.ctx, P, H, W, self.D, M, _p(a["means3D"]), _p(a["shs"]), _p(a["colors_precomp"]), _p(a["opacities"]),
            _p(a["scales"]), _p(a["rotations"]), _p(a["cov3D_precomp"]), ctypes.c_float(self.scale_modifier),
            _p(a["viewmatrix"]), _p(a["projmatrix"]), _p(a["campos"]), _p(a["bg"]), ctypes.c_float(self.tanfovx),
            ctypes.c_float(self.tanfovy), _p(color), _p(radii), _p(depth), _p(alpha))
        if rc != 0:
            raise ValueError("oracle_raster_forward: bad arguments (rc=%d)" % rc)
        self.alpha = alpha
        return color, radii, depth, alpha

    @property
    def num_rendered(self):
        return int(self.lib.oracle_num_rendered(self.ctx))

    def binning(self):
        """(keys u64 [R] = tile<<32|depth_bits, point_list u32 [R], ranges u32 [T,2], tiles_touched u32 [P],
        n_contrib u32 [H,W])"""
        R = self.num_rendered
        tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        keys = np.zeros((R,), np.uint64)
        vals = np.zeros((R,), np.uint32)
        ranges = np.zeros((tiles, 2), np.uint32)
        tt = np.zeros((self.P,), np.uint32)
        nc = np.zeros((self.H, self.W), np.uint32)
        self.lib.oracle_copy_binning(self.ctx, _p(keys), _p(vals), _p(ranges), _p(tt), _p(nc))
        return keys, vals, ranges, tt, nc

    def pixel_margins(self, pix_ids):
        """Knife-edge margin (see oracle_pixel_margins in raster_oracle.c) of the given flat pixel ids y * W + x."""
        ids = np.ascontiguousarray(np.asarray(pix_ids, dtype=np.int32))
        out = np.zeros(ids.shape[0], np.float32)
        if ids.shape[0]:
            self.lib.oracle_pixel_margins(self.ctx, _p(ids), ctypes.c_int(ids.shape[0]), _p(out))
        return out

    def knife_edge_gaussians(self, thresh=2e-5, downstream=False, sharing=False):
        """(bool [P] mask of the Gaussians that are the subject of a knife-edge threshold test, number of such pixels);
        with downstream=True also a [P] mask of the Gaussians blended behind such a subject at some pixel; with
        sharing=True instead a [P] mask of every Gaussian that blends at a knife-edge pixel, in front of the subject or
        behind it (raster_oracle.c: oracle_knife_edge_gaussians2)."""
        flags = np.zeros(self.P, np.uint8)
        down = np.zeros(self.P, np.uint8) if downstream else None
        share = np.zeros(self.P, np.uint8) if sharing else None
        self.lib.oracle_knife_edge_gaussians2.restype = ctypes.c_int
        n = self.lib.oracle_knife_edge_gaussians2(self.ctx, ctypes.c_float(thresh), _p(flags), _p(down), _p(share))
        if sharing:
            return flags.astype(bool), int(n), share.astype(bool)
        if downstream:
            return flags.astype(bool), int(n), down.astype(bool)
        return flags.astype(bool), int(n)

    def geom(self):
        P = self.P
        out = dict(means2D=np.zeros((P, 2), np.float32), depths=np.zeros((P,), np.float32),
                   cov3D=np.zeros((P, 6), np.float32), rgb=np.zeros((P, 3), np.float32),
                   conic_opacity=np.zeros((P, 4), np.float32), clamped=np.zeros((P, 3), np.uint8))
        self.lib.oracle_copy_geom(self.ctx, _p(out["means2D"]), _p(out["depths"]), _p(out["cov3D"]), _p(out["rgb"]),
                                  _p(out["conic_opacity"]), _p(out["clamped"]))
        return out

    def backward(self, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None, alpha_out=None):
        """`alpha_out` [1,H,W]: the forward's alpha image the backward derives T_final = 1 - alpha from (the fork passes
        its own forward output).  Default: this oracle's forward output.  Passing the alpha image of the implementation
        under test isolates the backward: where a pixel is nearly opaque, T_final is the difference of two nearly equal
        numbers and a 1e-7 difference between two forwards' alpha becomes a percent-level difference in every T_j of
        that pixel — a property of the reference's formulation, not of either implementation."""
        a = self.args
        P, M = self.P, self.M
        g = dict(means3D=np.zeros((P, 3), np.float32), means2D=np.zeros((P, 3), np.float32),
                 shs=np.zeros((P, max(M, 1), 3), np.float32) if a["shs"] is not None else None,
                 colors_precomp=np.zeros((P, 3), np.float32) if a["colors_precomp"] is not None else None,
                 opacities=np.zeros((P, 1), np.float32),
                 scales=np.zeros((P, 3), np.float32) if a["scales"] is not None else None,
                 rotations=np.zeros((P, 4), np.float32) if a["rotations"] is not None else None,
                 cov3D_precomp=np.zeros((P, 6), np.float32) if a["cov3D_precomp"] is not None else None)
        acc = np.zeros((P, 10), np.float64)
        gc, gd, ga = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
        rc = self.lib.oracle_raster_backward(
            self.ctx, _p(a["means3D"]), _p(a["shs"]), _p(a["colors_precomp"]), _p(a["scales"]), _p(a["rotations"]),
            _p(a["cov3D_precomp"]), ctypes.c_float(self.scale_modifier), _p(a["viewmatrix"]), _p(a["projmatrix"]),
            _p(a["campos"]), _p(a["bg"]), ctypes.c_float(self.tanfovx), ctypes.c_float(self.tanfovy),
            _p(self.alpha if alpha_out is None else _f32(alpha_out).reshape(self.alpha.shape)),
            _p(gc), _p(gd), _p(ga), _p(g["means3D"]), _p(g["means2D"]), _p(g["shs"]), _p(g["colors_precomp"]),
            _p(g["opacities"]), _p(g["scales"]), _p(g["rotations"]), _p(g["cov3D_precomp"]), _p(acc))
        if rc != 0:
            raise RuntimeError("oracle_raster_backward rc=%d" % rc)
        g["_acc"] = acc
        return g


def knn_mean_dist2(points):
    """Mean squared distance to the 3 nearest neighbours (simple_knn.cu:119-221 semantics), brute force O(P^2)."""
    global _knn
    if _knn is None:
        build()
        _knn = ctypes.CDLL(os.path.join(_DIR, "libknn_oracle.so"))
    pts = _f32(points)
    out = np.zeros((pts.shape[0],), np.float32)
    _knn.oracle_knn_mean_dist2(ctypes.c_int(pts.shape[0]), _p(pts), _p(out))
    return out
