import sys, os, time, torch, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
dev = torch.device("cuda")
P, H, W, V = 2000, 64, 64, 4
sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
cams = scenes.train_cameras(V, seed=42, H=H, W=W)
bg = torch.zeros(3, device=dev)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
       viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
       campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
gC = torch.randn((V, 3, H, W), device=dev) * 1e-3; gD = torch.randn((V, 1, H, W), device=dev) * 1e-3
plist = [t[n] for n in ["means3D", "shs", "opacities", "scales", "rotations"]]
def step():
    m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    color, radii, depth, alpha = rasterize_views(t["means3D"], m2d, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    return torch.autograd.grad([color, depth], plist + [m2d], [gC, gD])
for _ in range(50): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(500): step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print("tiny scene: host enqueue %.3f ms/step, wall %.3f ms/step" % (th / 500 * 1e3, ta / 500 * 1e3))
