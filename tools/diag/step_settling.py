"""Wall time of each of the first raster steps of a fresh process (what does a short --warmup leave in the timed region?)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
dev = torch.device("cuda")
P, H, W, V = 100000, 1024, 1024, 4
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
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step wall (sync each), ms:", " ".join("%.2f" % x for x in ts))
# batches of 20 without per-step sync
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); print("20 steps: %.4f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
