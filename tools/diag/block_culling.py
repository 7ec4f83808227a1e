"""How many (list entry, 8x4 pixel block) pairs survive different culling tests on the bench scene (CPU, oracle).

render_forward stages every list entry with an 8-bit block mask; a pair outside the alpha >= 1/255 ellipse cannot
contribute.  Compares: (a) the ellipse's axis-aligned extent (the shipped test), (b) (a) AND lambda_min * dist^2 <= t,
(c) the exact minimum of the quadratic form over the block, (d) truth: some pixel of the block passes the alpha test."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from oracle.oracle import RasterOracle

P, H, W = int(os.environ.get("P", 100000)), 1024, 1024
look = os.environ.get("LOOK", "init")
sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
if look == "trained":
    sc = scenes.trained_look(sc, seed=1) if hasattr(scenes, "trained_look") else sc
cam = scenes.train_cameras(4, seed=42, H=H, W=W)[0]
ro = RasterOracle()
ro.forward(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=np.zeros(3, np.float32),
           scale_modifier=1.0, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], sh_degree=0, campos=cam["campos"],
           means3D=sc["means3D"], shs=sc["shs"], opacities=sc["opacities"], scales=sc["scales"], rotations=sc["rotations"])
g = ro.geom()
keys, vals, ranges, tt, nc = ro.binning()
tiles_x = W // 16
tile_of = (keys >> np.uint64(32)).astype(np.int64)
gi = vals.astype(np.int64)
tx0 = (tile_of % tiles_x) * 16.0
ty0 = (tile_of // tiles_x) * 16.0
cx, cy = g["means2D"][gi, 0] - tx0, g["means2D"][gi, 1] - ty0
A, B, C, o = (g["conic_opacity"][gi, k].astype(np.float64) for k in range(4))
t2 = 2.0 * np.log(255.0 * o) + 0.02
det = A * C - B * B
inv = t2 / det
hx = np.sqrt(np.maximum(inv * C, 0)) * 1.01 + 0.05
hy = np.sqrt(np.maximum(inv * A, 0)) * 1.01 + 0.05
lam_min = 0.5 * (A + C) - np.sqrt(0.25 * (A - C) ** 2 + B * B)
R = len(gi)
tot = dict(aabb=0, circle=0, exact=0, truth=0)
for by in range(4):
    for bx in range(2):
        x0, x1, y0, y1 = 8.0 * bx, 8.0 * bx + 7, 4.0 * by, 4.0 * by + 3
        aabb = (t2 > 0) & (cx - hx <= x1) & (cx + hx >= x0) & (cy - hy <= y1) & (cy + hy >= y0)
        ddx = np.maximum(np.maximum(x0 - cx, cx - x1), 0)
        ddy = np.maximum(np.maximum(y0 - cy, cy - y1), 0)
        circ = aabb & (lam_min * (ddx * ddx + ddy * ddy) <= t2)
        # exact min of q(dx,dy) = A dx^2 + 2 B dx dy + C dy^2 over the block (coordinates relative to the centre)
        X0, X1, Y0, Y1 = x0 - cx, x1 - cx, y0 - cy, y1 - cy
        q = lambda x, y: A * x * x + 2 * B * x * y + C * y * y
        inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
        cand = []
        for xe in (X0, X1):
            ys = np.clip(-B * xe / C, Y0, Y1); cand.append(q(xe, ys))
        for ye in (Y0, Y1):
            xs = np.clip(-B * ye / A, X0, X1); cand.append(q(xs, ye))
        qmin = np.where(inside, 0.0, np.minimum.reduce(cand))
        exact = (t2 > 0) & (qmin <= t2)
        truth = np.zeros(R, bool)
        for yy in range(4):
            for xx in range(8):
                dx, dy = X0 + xx, Y0 + yy
                power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
                al = np.minimum(0.99, o * np.exp(power))
                truth |= (power <= 0) & (al >= 1.0 / 255.0)
        assert not (truth & ~exact).any() and not (truth & ~circ).any()
        tot["aabb"] += int(aabb.sum()); tot["circle"] += int(circ.sum()); tot["exact"] += int(exact.sum()); tot["truth"] += int(truth.sum())
print("instances %d  (entry, block) pairs: all %d" % (R, 8 * R))
for k, v in tot.items():
    print("  %-7s %9d  %.3f of aabb  (%.2f blocks / entry)" % (k, v, v / tot["aabb"], v / R))

# ---- backward: 8x8 quadrants (one wave pass = 64 pixels per touched quadrant)
def count(rects, label):
    tq = dict(aabb=0, circle=0, exact=0)
    for (x0, x1, y0, y1) in rects:
        aabb = (t2 > 0) & (cx - hx <= x1) & (cx + hx >= x0) & (cy - hy <= y1) & (cy + hy >= y0)
        ddx = np.maximum(np.maximum(x0 - cx, cx - x1), 0)
        ddy = np.maximum(np.maximum(y0 - cy, cy - y1), 0)
        circ = aabb & (lam_min * (ddx * ddx + ddy * ddy) <= t2)
        X0, X1, Y0, Y1 = x0 - cx, x1 - cx, y0 - cy, y1 - cy
        q = lambda x, y: A * x * x + 2 * B * x * y + C * y * y
        inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
        cand = []
        for xe in (X0, X1):
            ys = np.clip(-B * xe / C, Y0, Y1); cand.append(q(xe, ys))
        for ye in (Y0, Y1):
            xs = np.clip(-B * ye / A, X0, X1); cand.append(q(xs, ye))
        qmin = np.where(inside, 0.0, np.minimum.reduce(cand))
        exact = (t2 > 0) & (qmin <= t2)
        tq["aabb"] += int(aabb.sum()); tq["circle"] += int(circ.sum()); tq["exact"] += int(exact.sum())
    npx = (rects[0][1] - rects[0][0] + 1) * (rects[0][3] - rects[0][2] + 1)
    print("%s (%d px per pass):" % (label, npx))
    for k, v in tq.items():
        print("  %-7s %9d passes  %.2f / entry   pixel-pairs %.1f / entry" % (k, v, v / R, v * npx / R))
count([(8.0 * bx, 8.0 * bx + 7, 8.0 * by, 8.0 * by + 7) for by in range(2) for bx in range(2)], "bwd quadrants 8x8")
count([(0.0, 15.0, 4.0 * by, 4.0 * by + 3) for by in range(4)], "bands 16x4")
count([(8.0 * bx, 8.0 * bx + 7, 4.0 * by, 4.0 * by + 3) for by in range(4) for bx in range(2)], "blocks 8x4")
alive = (t2 > 0)
print("entries with t2 > 0: %.3f" % alive.mean())

# ---- tile-level: instances under the reference rect / the tight extent rect / the exact tile-ellipse test
P_ = g["means2D"].shape[0]
mx, my = g["means2D"][:, 0].astype(np.float64), g["means2D"][:, 1].astype(np.float64)
A_, B_, C_, o_ = (g["conic_opacity"][:, k].astype(np.float64) for k in range(4))
vis = tt > 0
t2_ = 2.0 * np.log(np.maximum(255.0 * o_, 1e-30)) + 0.02
det_ = A_ * C_ - B_ * B_
hx_ = np.sqrt(np.maximum(t2_ / det_ * C_, 0)) * 1.01 + 0.05
hy_ = np.sqrt(np.maximum(t2_ / det_ * A_, 0)) * 1.01 + 0.05
# the reference radius from the conic's inverse (cov2D = inverse of the conic matrix)
ca, cb, cc = C_ / det_, -B_ / det_, A_ / det_
mid = 0.5 * (ca + cc)
lam1 = mid + np.sqrt(np.maximum(0.1, mid * mid - (ca * cc - cb * cb)))
rad = np.ceil(3.0 * np.sqrt(lam1))
def rect(rx, ry):
    x0 = np.clip(((mx - rx) / 16).astype(np.int64), 0, tiles_x); y0 = np.clip(((my - ry) / 16).astype(np.int64), 0, H // 16)
    x1 = np.clip(((mx + rx + 15) / 16).astype(np.int64), 0, tiles_x); y1 = np.clip(((my + ry + 15) / 16).astype(np.int64), 0, H // 16)
    return x0, y0, x1, y1
x0, y0, x1, y1 = rect(rad, rad)
ref_n = ((x1 - x0) * (y1 - y0))[vis].sum()
# tight: tiles whose pixel range [16 t, 16 t + 15] meets [m - h, m + h], inside the reference rect
tx0 = np.maximum(x0, np.floor((mx - hx_) / 16).astype(np.int64)); tx1 = np.minimum(x1, np.floor((mx + hx_) / 16).astype(np.int64) + 1)
ty0 = np.maximum(y0, np.floor((my - hy_) / 16).astype(np.int64)); ty1 = np.minimum(y1, np.floor((my + hy_) / 16).astype(np.int64) + 1)
tight_n = (np.maximum(tx1 - tx0, 0) * np.maximum(ty1 - ty0, 0))[vis & (t2_ > 0)].sum()
print("instances: reference rect %d (oracle %d), tight extent rect %d (%.3f)" % (ref_n, R, tight_n, tight_n / ref_n))
# exact: per instance of the oracle list, min of q over the tile's pixel rectangle
X0, X1, Y0, Y1 = 0.0 - cx, 15.0 - cx, 0.0 - cy, 15.0 - cy
q = lambda x, y: A * x * x + 2 * B * x * y + C * y * y
inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
cand = []
for xe in (X0, X1):
    ys = np.clip(-B * xe / C, Y0, Y1); cand.append(q(xe, ys))
for ye in (Y0, Y1):
    xs = np.clip(-B * ye / A, X0, X1); cand.append(q(xs, ye))
qmin = np.where(inside, 0.0, np.minimum.reduce(cand))
print("           exact tile / ellipse test %d (%.3f)" % (int(((t2 > 0) & (qmin <= t2)).sum()), ((t2 > 0) & (qmin <= t2)).sum() / R))
