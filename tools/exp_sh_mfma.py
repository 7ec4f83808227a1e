#!/usr/bin/env python3
"""SH colour contraction: scalar chain (GIP_RASTER_SH_SCALAR=1) against the matrix-core path (csrc/sh_mfma.hip), per-stage times
of the rasterizer at sh_degree 1..3 — 100k Gaussians, 1024^2, the 4-view launch set (and 1M Gaussians x 12 views at degree 3).
The SH kernels run inside the `preprocess` (forward) and `gather_bwd` (backward) stage brackets.  Prints one line per case;
profiles/r06_sh_mfma.txt is its output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import scenes
    from gaussianip_amd import GaussianRasterizationSettings
    from gaussianip_amd import rasterizer as R
    dev = torch.device("cuda")
    H = W = 1024
    for P, V, degs in ((100000, 4, (1, 2, 3)), (1000000, 12, (3,))):
        for deg in degs:
            sc = scenes.make_scene("human", P, seed=42, sh_degree=deg)
            rng = np.random.default_rng(1)
            sc["shs"][:, 0, :] = ((rng.uniform(0.2, 0.9, (P, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
            sc["shs"][:, 1:, :] = (rng.normal(size=(P, (deg + 1) ** 2 - 1, 3)) * 0.15).astype(np.float32)
            if P > 100000:
                sc["scales"] = (sc["scales"] / 1.6).astype(np.float32)
                sc["opacities"][:] = 0.6
            t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
            bg = torch.zeros(3, device=dev)
            cams = scenes.train_cameras(V, 42, H, W) if V == 4 else [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(V)]
            sts = [GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
                viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
                sh_degree=deg, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
            gC = torch.randn((V, 3, H, W), device=dev) * 1e-3
            res = {}
            for rep in range(2):
                for mode in ("scalar", "mfma"):
                    os.environ["GIP_RASTER_SH_SCALAR"] = "1" if mode == "scalar" else "0"
                    ms, nr = R.profile_stages(t["means3D"], t["opacities"], sts, gC, shs=t["shs"], scales=t["scales"],
                                              rotations=t["rotations"], iters=10)
                    res[mode] = ms
            print("P %7d V %2d sh_degree %d | preprocess stage: scalar %.4f ms  matrix cores %.4f ms | gather_bwd stage: scalar %.4f ms  "
                  "matrix cores %.4f ms | fwd+bwd sum of stages: scalar %.4f  matrix cores %.4f" % (
                      P, V, deg, res["scalar"]["preprocess"], res["mfma"]["preprocess"], res["scalar"]["gather_bwd"], res["mfma"]["gather_bwd"],
                      sum(res["scalar"].values()), sum(res["mfma"].values())), flush=True)


if __name__ == "__main__":
    main()
