import os, sys, torch, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings
from gaussianip_amd import rasterizer as R
dev = torch.device("cuda"); P, H, W, V = 100000, 1024, 1024, 4
sc = scenes.make_scene("human", P, seed=42, sh_degree=0); cams = scenes.train_cameras(V, seed=42, H=H, W=W)
bg = torch.zeros(3, device=dev)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
       viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
       campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
for _ in range(5):
    (outs, plan) = R.forward_with_state(t["means3D"], t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    torch.cuda.synchronize()
    h = R.state_views(plan)["header"].cpu().numpy()
    print("header words", h[:16], "ticks (10 ns): %d %d %d" % (h[13], h[14], h[15]))
