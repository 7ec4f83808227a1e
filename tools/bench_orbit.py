#!/usr/bin/env python3
"""BASELINE.json configs[4], render part: 1M Gaussians (post-densify look: the 100k init split / cloned up to 1M, scales
/1.6 per split as gaussian_model.py:371, opacity 0.6), 1024^2, 36-view orbit (elevation 5, distance 1.8, fovy 70;
configs/exp.yaml:37-40), forward only (no_grad), plus the same views forward+backward.  Views go through the
rasterizer in launch sets of up to 12.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import scenes
    from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
    from gaussianip_amd import rasterizer as R
    dev = torch.device("cuda")
    P, H, W, NV, SET = 1000000, 1024, 1024, 36, 12
    sc = scenes.make_scene("human", P, seed=42)
    sc["scales"] = (sc["scales"] / 1.6).astype(np.float32)
    sc["opacities"][:] = 0.6
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
    bg = torch.zeros(3, device=dev)
    cams = [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(NV)]
    sts = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
        sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]

    def orbit(grad):
        tot = 0.0
        for s in range(0, NV, SET):
            tt = {k: v.requires_grad_(grad) for k, v in t.items()} if grad else t
            color, radii, depth, alpha = rasterize_views(tt["means3D"], None, tt["opacities"], sts[s:s + SET], shs=tt["shs"],
                                                         scales=tt["scales"], rotations=tt["rotations"])
            if grad:
                (color.sum() + depth.sum()).backward()
                for v in t.values():
                    v.grad = None
        return tot

    res = {}
    for name, grad in (("forward", False), ("forward_backward", True)):
        ctx = torch.no_grad() if not grad else torch.enable_grad()
        with ctx:
            orbit(grad)
            orbit(grad)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                orbit(grad)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        res[name] = {"ms_per_orbit": round(dt * 1e3, 2), "views_per_s": round(NV / dt, 1), "mpix_per_s": round(NV * H * W / dt / 1e6, 1)}
    with torch.no_grad():
        (_, plan) = R.forward_with_state(t["means3D"], t["opacities"], sts[:1], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        hdr = R.state_views(plan)["header"].cpu().numpy()
    print(json.dumps({"workload": "configs[4] render part: 1M Gaussians, 1024^2, 36-view orbit", "gaussians": P, "views": NV,
                      "num_rendered_view0": int(hdr[1]), "max_tile_list": int(hdr[3]), **res}), flush=True)


if __name__ == "__main__":
    main()
