"""Renders with and without GipRasterConfig::forward_only (the kernel then skips the per-segment checkpoints and the
n_contrib / final_T images): GPU time of the 4-view launch set by HIP events, 100k Gaussians at 1024^2 — the init state of the
headline bench and the trained-looking state (longer lists: more checkpoints).
usage: forward_only_timing.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from gaussianip_amd import rasterizer as rz  # noqa: E402
from gaussianip_amd.rasterizer import GaussianRasterizationSettings  # noqa: E402

dev = torch.device("cuda")
P, H, W, V = 100000, 1024, 1024, 4
bg = torch.zeros(3, device=dev)
cams = scenes.train_cameras(V, seed=42, H=H, W=W)
sts = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
                                     viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
                                     sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
for state in ("init", "trained"):
    sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
    if state == "trained":
        sc = scenes.trained_look(sc, seed=7)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
    for fo in (False, True, False, True):
        plan = rz._build_plan(t["means3D"], t["shs"], None, t["opacities"], t["scales"], t["rotations"], None, sts)
        rz._forward_with_policy(plan, False, forward_only=fo)           # sizes the capacity
        cap = plan.capacity
        for _ in range(5):
            rz._run_forward(plan, cap, fo)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            rz._run_forward(plan, cap, fo)
        b.record()
        torch.cuda.synchronize()
        print("%-8s forward_only=%d  %.4f ms per 4-view forward" % (state, fo, a.elapsed_time(b) / 50), flush=True)
