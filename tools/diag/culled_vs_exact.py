"""Default (culled) mode against exact_lists mode on one adversarial scene: per-tensor differences and the worst Gaussian."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from gaussianip_amd import GaussianRasterizationSettings
from gaussianip_amd import rasterizer as R
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(100 + seed)
P, H, W = 6000, 160, 208
sc = scenes.make_scene("stress", P, seed=seed, sh_degree=1)
sc["scales"] = (sc["scales"] * np.exp(rng.uniform(-2.5, 2.5, (P, 3)))).astype(np.float32)
op = rng.uniform(0.0, 1.0, (P, 1)); near = rng.random((P, 1)) < 0.3
op[near] = (1.0 / 255.0) * np.exp(rng.uniform(-0.2, 0.6, int(near.sum())))
if len(sys.argv) > 2 and sys.argv[2] == "translucent":
    op[~near] *= 0.004
sc["opacities"] = op.astype(np.float32)
cam = scenes.camera(12.0, 40.0 + 30.0 * seed, 0.9 + 0.4 * seed, 65.0, H, W)
dev = "cuda"
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.tensor([0.2, 0.1, 0.3], device=dev),
     scale_modifier=1.0, viewmatrix=torch.from_numpy(cam["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(cam["projmatrix"]).to(dev), sh_degree=1,
     campos=torch.from_numpy(cam["campos"]).to(dev), prefiltered=False, debug=False)
g = torch.Generator(device=dev).manual_seed(seed)
gC = torch.randn(1, 3, H, W, device=dev, generator=g); gD = torch.randn(1, 1, H, W, device=dev, generator=g); gA = torch.randn(1, 1, H, W, device=dev, generator=g)
outs = {}
for mode in ("1", "0"):
    os.environ["GIP_RASTER_EXACT_LISTS"] = mode
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(1, P, 3, device=dev, requires_grad=True)
    color, radii, depth, alpha = R.rasterize_views(t["means3D"], m2d, t["opacities"], [st], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    grads = torch.autograd.grad([color, depth, alpha], [t["means3D"], t["shs"], t["opacities"], t["scales"], t["rotations"], m2d], [gC, gD, gA])
    outs[mode] = (color.detach(), depth.detach(), alpha.detach(), radii, grads)
names = ["means3D", "shs", "opacities", "scales", "rotations", "means2D"]
ce, de, ae, re_, ge = outs["1"]; cc, dc, ac, rc, gc = outs["0"]
print("images: color %.2e depth %.2e alpha %.2e  max alpha %.6f  pixels with alpha > 0.999: %d" % (float((ce-cc).abs().max()), float((de-dc).abs().max()), float((ae-ac).abs().max()), float(ae.max()), int((ae > 0.999).sum())))
for n, a, b in zip(names, ge, gc):
    a2, b2 = a.reshape(P, -1), b.reshape(P, -1)
    d = (a2 - b2).abs().max(dim=1).values
    i = int(d.argmax())
    print("%-10s max|exact| %.3e  max diff %.3e (%.2e of max) at Gaussian %d: exact %s culled %s  opacity %.5f scales %s radius %d" % (
        n, float(a.abs().max()), float(d.max()), float(d.max() / a.abs().max()), i, a2[i].tolist()[:3], b2[i].tolist()[:3], float(sc["opacities"][i]), sc["scales"][i].tolist(), int(re_[0, i])))
