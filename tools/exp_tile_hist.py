import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings
from gaussianip_amd import rasterizer as R
dev = torch.device("cuda")
P, H, W, V = int(os.environ.get("P", 100000)), 1024, 1024, 4
sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
cams = scenes.train_cameras(V, seed=42, H=H, W=W)
if os.environ.get("ORBIT"):      # the configs[4] scene of tools/exp_orbit_stages.py / bench_orbit.py
    sc["scales"] = (sc["scales"] / 1.6).astype(np.float32); sc["opacities"][:] = 0.6
    cams = [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(V)]
bg = torch.zeros(3, device=dev)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
       viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
       campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
(outs, plan) = R.forward_with_state(t["means3D"], t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
sv = R.state_views(plan)
ts = sv["tile_start"].cpu().numpy().astype(np.int64)
n = np.diff(ts)
nc = sv["n_contrib"].cpu().numpy()
print("tiles", n.size, "occupied", int((n > 0).sum()), "sum", int(n.sum()))
occ = n[n > 0]
print("mean %.0f p50 %d p90 %d p99 %d max %d" % (occ.mean(), np.percentile(occ, 50), np.percentile(occ, 90), np.percentile(occ, 99), occ.max()))
for thr in (256, 512, 1024, 2048, 4096, 8192, 16384):
    print("tiles > %4d: %5d holding %.1f%% of entries" % (thr, int((n > thr).sum()), 100.0 * n[n > thr].sum() / n.sum()))
# how deep do pixels actually walk (n_contrib = last contributing entry): early termination
print("n_contrib: mean over covered px %.0f, max %d" % (nc[nc > 0].mean(), nc.max()))
