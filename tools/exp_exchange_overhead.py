"""Single-GPU estimate of what the N>1 raster step adds besides the collectives themselves: the bench step with the
exchange packing / unpacking kernels, all_reduce stubbed out."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings, parallel
from gaussianip_amd.renderer import rasterize_views
import torch.distributed as dist
dev = torch.device("cuda"); P, V, H, W = 100000, 4, 1024, 1024
sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
cams = scenes.train_cameras(V, seed=42, H=H, W=W)
bg = torch.zeros(3, device=dev)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev), sh_degree=0,
        campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
gC = torch.randn((V, 3, H, W), device=dev) * 1e-3; gD = torch.randn((V, 1, H, W), device=dev) * 1e-3
plist = [t[n] for n in ["means3D", "shs", "opacities", "scales", "rotations"]]
class W_:
    def wait(self): pass
MODE = os.environ.get("MODE", "both")
def step(multi):
    m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    color, radii, depth, alpha = rasterize_views(t["means3D"], m2d, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    if multi and MODE in ("both", "max"):
        pending = parallel.exchange_forward_stats(radii, depth)
    grads = torch.autograd.grad([color, depth], plist + [m2d], [gC, gD])
    if multi:
        for p_, g_ in zip(plist, grads[:-1]): p_.grad = g_
        if MODE in ("both", "sum"): parallel.exchange_sum(plist, viewspace_grads=grads[-1])
        if MODE in ("both", "max"): pending.wait()
def timeit(multi, n=200):
    for _ in range(20): step(multi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(multi)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
parallel._on = lambda group=None: True
dist.all_reduce = lambda *a, **k: W_()
dist.get_world_size = lambda group=None: 2
for _ in range(3):
    print("N=1 step %.4f ms | N>1 step without the collectives %.4f ms" % (timeit(False), timeit(True)), flush=True)
