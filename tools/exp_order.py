"""Stage times for the bench scene in its generated (limb-coherent) order and in a random order."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings
from gaussianip_amd import rasterizer as R
dev = torch.device("cuda")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
H = W = 1024; V = 4
sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
cams = scenes.train_cameras(V, seed=42, H=H, W=W)
bg = torch.zeros(3, device=dev)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
       viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
       campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
gC = torch.randn((V, 3, H, W), device=dev) * 1e-3; gD = torch.randn((V, 1, H, W), device=dev) * 1e-3
perm = np.random.default_rng(0).permutation(P)
for name, order in (("generated order", None), ("random order", perm)):
    t = {k: torch.from_numpy(v if order is None else v[order]).to(dev) for k, v in sc.items()}
    stages, nr = R.profile_stages(t["means3D"], t["opacities"], sts, gC, gD, None, shs=t["shs"], scales=t["scales"], rotations=t["rotations"], iters=10)
    print(name, nr, {k: round(v, 4) for k, v in stages.items()}, "sum %.4f" % sum(stages.values()), flush=True)
