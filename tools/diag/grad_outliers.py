"""Where do the largest gradient errors against the oracle sit?  (diagnostic, GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from oracle import oracle as orc
from test_gpu_raster_parity import _settings, _dev, _oracle_forward
from gaussianip_amd import GaussianRasterizer
from gaussianip_amd import rasterizer as R

H = W = 1024; P = 100000
orc.build(); orc.set_threads(orc.max_threads())
sc = scenes.make_scene("human", P, seed=42)
cam = scenes.train_cameras(4, 42, H, W)[0]
bg = (0.0, 0.0, 0.0)
rng = np.random.default_rng(3)
gC = rng.normal(size=(3, H, W)).astype(np.float32); gD = rng.normal(size=(1, H, W)).astype(np.float32); gA = rng.normal(size=(1, H, W)).astype(np.float32)
for variant in ("all", "color_only"):
    gd, ga = (gD, gA) if variant == "all" else (None, None)
    ro, out = _oracle_forward(orc, sc, cam, H, W, bg, 0)
    go = ro.backward(gC, gd, ga)
    st = _settings(cam, H, W, bg, 0)
    t = {k: _dev(v).requires_grad_(True) for k, v in sc.items()}
    m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
    color, radii, depth, alpha = GaussianRasterizer(st)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    loss = (color * _dev(gC)).sum()
    if gd is not None:
        loss = loss + (depth * _dev(gD)).sum() + (alpha * _dev(gA)).sum()
    loss.backward(); torch.cuda.synchronize()
    (o2, plan) = R.forward_with_state(t["means3D"].detach(), t["opacities"].detach(), [st], shs=t["shs"].detach(), scales=t["scales"].detach(), rotations=t["rotations"].detach())
    sv = R.state_views(plan)
    keys, vals, ranges, tt, nc = ro.binning()
    mynx = sv["n_contrib"][0].cpu().numpy().astype(np.uint32)
    mism = np.argwhere(mynx != nc)
    print(variant, "n_contrib mismatching pixels:", len(mism))
    for (y, x) in mism[:10]:
        print("   pixel", x, y, "ours", mynx[y, x], "oracle", nc[y, x], "alpha", float(alpha[0, y, x]), "g", gC[:, y, x])
    ours = t["shs"].grad.cpu().numpy().reshape(P, 3).astype(np.float64); ref = go["shs"].reshape(P, 3).astype(np.float64)
    err = np.abs(ours - ref); top = np.abs(ref).max()
    print(" shs top", top, "max err", err.max(), "median err", np.median(err), "p99.9", np.quantile(err, 0.999))
    worst = np.argsort(-err.max(1))[:8]
    geo = ro.geom()
    for g in worst:
        e3 = np.abs(t["means3D"].grad[g].cpu().numpy() - go["means3D"][g]).max() / (np.abs(go["means3D"]).max())
        print("  g", g, "err", err[g], "ref", ref[g], "radius", int(radii[g]), "tiles", tt[g], "depth", geo["depths"][g], "xy", geo["means2D"][g], "means3D relerr", e3)
    # histogram of errors
    print(" count err>1e-5*top:", int((err > 1e-5 * top).sum()), " >1e-4*top:", int((err > 1e-4 * top).sum()))
