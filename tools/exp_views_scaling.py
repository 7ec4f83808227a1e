"""Stage times of the raster step vs the number of views per launch set (is a stage throughput- or critical-path-bound?)."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings
from gaussianip_amd import rasterizer as R
dev = torch.device("cuda"); P, H, W = 100000, 1024, 1024
sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
bg = torch.zeros(3, device=dev)
allc = scenes.train_cameras(4, seed=42, H=H, W=W)
for V in (1, 2, 4, 8):
    cams = [allc[i % 4] for i in range(V)]
    sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
           viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
           campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
    gC = torch.randn((V, 3, H, W), device=dev) * 1e-3; gD = torch.randn((V, 1, H, W), device=dev) * 1e-3
    stages, nr = R.profile_stages(t["means3D"], t["opacities"], sts, gC, gD, None, shs=t["shs"], scales=t["scales"], rotations=t["rotations"], iters=20)
    print("V=%d" % V, {k: round(v, 4) for k, v in stages.items()}, "sum %.4f  per view %.4f" % (sum(stages.values()), sum(stages.values()) / V))
