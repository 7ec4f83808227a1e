#!/usr/bin/env python3
"""Where does the means2D error of view 1 of the headline 4-view set come from (max-normalised 4.2e-4
against 3.5e-7 for views 0, 2, 3)?  Renders each view alone (default and exact-lists mode), compares every gradient with
the oracle given the same alpha image, and prints the rows with the largest absolute error together with what the
oracle knows about them (depth, radius, opacity, knife-edge margins of the pixels they touch)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_gpu_raster_parity import _dev, _oracle_forward, _settings  # noqa: E402
from gaussianip_amd import GaussianRasterizer  # noqa: E402

H = W = 1024
P = 100000
orc.build()
orc.set_threads(orc.max_threads())
sc = scenes.make_scene("human", P, seed=42)
cams = scenes.train_cameras(4, 42, H, W)
rng = np.random.default_rng(5)
gC, gD, gA = rng.normal(size=(4, 3, H, W)).astype(np.float32), rng.normal(size=(4, 1, H, W)).astype(np.float32), rng.normal(size=(4, 1, H, W)).astype(np.float32)
for v in (0, 1):
    for exact in ("0", "1"):
        os.environ["GIP_RASTER_EXACT_LISTS"] = exact
        st = _settings(cams[v], H, W, (0.0, 0.0, 0.0), 0)
        t = {k: _dev(a).requires_grad_(True) for k, a in sc.items()}
        m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(st)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"],
                                                            scales=t["scales"], rotations=t["rotations"])
        ((color * _dev(gC[v])).sum() + (depth * _dev(gD[v])).sum() + (alpha * _dev(gA[v])).sum()).backward()
        torch.cuda.synchronize()
        ro, (o_color, o_radii, o_depth, o_alpha) = _oracle_forward(orc, sc, cams[v], H, W, (0.0, 0.0, 0.0), 0)
        go = ro.backward(gC[v], gD[v], gA[v], alpha_out=alpha.detach().cpu().numpy())
        ours = m2.grad.cpu().numpy().astype(np.float64)
        ref = go["means2D"].astype(np.float64)
        err = np.abs(ours - ref).max(axis=1)
        top = np.abs(ref).max()
        order = np.argsort(-err)[:6]
        geom = ro.geom()
        print("view %d exact_lists=%s: means2D max-normalised %.2e; alpha max |diff| vs oracle %.2e; colour %.2e" % (
            v, exact, err.max() / top, np.abs(alpha.detach().cpu().numpy() - o_alpha).max(), np.abs(color.detach().cpu().numpy() - o_color).max()))
        for r in order:
            print("   row %6d  err %.3e  ref (%.4e, %.4e)  ours (%.4e, %.4e)  depth %.4f radius %d  mean2D (%.1f, %.1f) opacity*  %.3f" % (
                r, err[r], ref[r, 0], ref[r, 1], ours[r, 0], ours[r, 1], geom["depths"][r], o_radii[r], geom["means2D"][r, 0], geom["means2D"][r, 1],
                geom["conic_opacity"][r, 3]))
        # pixel-level: where do the two alpha images differ most (same list walk?)
        da = np.abs(alpha.detach().cpu().numpy()[0] - o_alpha[0])
        iy, ix = np.unravel_index(int(da.argmax()), da.shape)
        print("   largest alpha difference %.3e at pixel (%d, %d), oracle alpha %.6f, margin %s" % (
            da.max(), ix, iy, o_alpha[0, iy, ix], ro.pixel_margins(np.array([iy * W + ix], np.int32))))
os.environ["GIP_RASTER_EXACT_LISTS"] = "0"
