"""Which Gaussian carries the largest element-wise colour-gradient error in the 4-view headline comparison, and how
close to a threshold flip is it?  (diagnostic, GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from oracle import oracle as orc
from test_gpu_raster_parity import _settings, _dev, _oracle_forward
import test_gpu_headline_parity as T
from gaussianip_amd import rasterize_views
H = W = 1024; P = 100000
orc.build(); orc.set_threads(orc.max_threads())
sc = T._look("init"); cams = scenes.train_cameras(4, 42, H, W); bg = (0.0, 0.0, 0.0)
gC, gD, gA = T._upstream(5, V=4)
sts = [_settings(c, H, W, bg, 0) for c in cams]
t = {k: _dev(v).requires_grad_(True) for k, v in sc.items()}
m2 = torch.zeros(4, P, 3, device="cuda", requires_grad=True)
color, radii, depth, alpha = rasterize_views(t["means3D"], m2, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
((color * _dev(gC)).sum() + (depth * _dev(gD)).sum() + (alpha * _dev(gA)).sum()).backward()
torch.cuda.synchronize()
alpha_np = alpha.detach().cpu().numpy()
ros, grads = [], []
for v, cam in enumerate(cams):
    ro, out = _oracle_forward(orc, sc, cam, H, W, bg, 0)
    ros.append(ro); grads.append(ro.backward(gC[v], gD[v], gA[v], alpha_out=alpha_np[v]))
tot = sum(g["shs"].astype(np.float64) for g in grads).reshape(P, 3)
ours = t["shs"].grad.cpu().numpy().reshape(P, 3).astype(np.float64)
top = np.abs(tot).max(); err = np.abs(ours - tot); big = np.abs(tot) > 1e-3 * top
rel = np.where(big, err / np.maximum(np.abs(tot), 1e-30), 0)
for thr in (2e-5, 1e-4, 1e-3, 1e-2):
    kn = np.logical_or.reduce([ro.knife_edge_gaussians(thr)[0] for ro in ros])
    r2 = rel.copy(); r2[kn] = 0
    g = int(np.argmax(r2.max(1)))
    print("thresh %.0e: flagged %d, worst rel %.3e at g=%d (tot %s err %s)" % (thr, kn.sum(), r2.max(), g, tot[g], err[g]))
g = int(np.argmax(rel.max(1)))
print("worst overall g", g, rel[g], "per-view contributions", [gr["shs"].reshape(P, 3)[g] for gr in grads])
print("per-view |ours_v - ref_v| not available (summed); radii", [int(radii[v, g]) for v in range(4)])
