#!/usr/bin/env python3
"""BASELINE.json configs[4] launch set for the profiler: 1M Gaussians, 1024^2, ONE 12-view set of the orbit, forward + backward,
`n` times (argv[1], default 3) after a sizing call; argv[2] = views per launch set (default 12: how the record tables of a set
compete for the caches is what round 6 measured with it).  Used by tools/collect_profiles.sh (kernel stats + HBM counters at 1M)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import scenes
    from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda")
    P, H, W, V = 1000000, 1024, 1024, (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    sc = scenes.make_scene("human", P, seed=42)
    sc["scales"] = (sc["scales"] / 1.6).astype(np.float32)
    sc["opacities"][:] = 0.6
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
    bg = torch.zeros(3, device=dev)
    cams = [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(V)]
    sts = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
        sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
    gen = torch.Generator(device=dev).manual_seed(4321)
    gC = torch.randn((V, 3, H, W), device=dev, generator=gen) * 1e-3
    gD = torch.randn((V, 1, H, W), device=dev, generator=gen) * 1e-3
    names = ["means3D", "shs", "opacities", "scales", "rotations"]
    for _ in range(n + 1):
        color, radii, depth, alpha = rasterize_views(t["means3D"], None, t["opacities"], sts, shs=t["shs"], scales=t["scales"],
                                                     rotations=t["rotations"])
        torch.autograd.grad([color, depth], [t[k] for k in names], [gC, gD])
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
