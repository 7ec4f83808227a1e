import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings
from gaussianip_amd import rasterizer as R
dev = torch.device("cuda")
P, H, W, V = 1000000, 1024, 1024, 12
sc = scenes.make_scene("human", P, seed=42)
sc["scales"] = (sc["scales"] / 1.6).astype(np.float32); sc["opacities"][:] = 0.6
cams = [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(V)]
bg = torch.zeros(3, device=dev)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
       viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
       campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
gC = torch.randn((V, 3, H, W), device=dev) * 1e-3; gD = torch.randn((V, 1, H, W), device=dev) * 1e-3
stages, nr = R.profile_stages(t["means3D"], t["opacities"], sts, gC, gD, None, shs=t["shs"], scales=t["scales"], rotations=t["rotations"], iters=3)
print("num_rendered", nr, {k: round(v, 3) for k, v in stages.items()})
