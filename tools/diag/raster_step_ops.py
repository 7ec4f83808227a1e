#!/usr/bin/env python3
"""Which PyTorch ops launch the non-rasterizer kernels (fills, copies) of bench.py's raster step?  torch.profiler over a few
steady-state steps of the same step function, grouped by op and Python stack."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from torch.profiler import ProfilerActivity, profile
    import scenes
    from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
    dev = torch.device("cuda")
    P, H, W, V = 100000, 1024, 1024, 4
    sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
    bg = torch.zeros(3, device=dev)
    cams = scenes.train_cameras(V, seed=42, H=H, W=W)
    sts = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
        sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
    gen = torch.Generator(device=dev).manual_seed(1234)
    gC = torch.randn((V, 3, H, W), device=dev, generator=gen) * 1e-3
    gD = torch.randn((V, 1, H, W), device=dev, generator=gen) * 1e-3
    plist = [t[n] for n in ["means3D", "shs", "opacities", "scales", "rotations"]]
    zeros = torch.zeros((V, P, 3), device=dev)

    def step():
        m2d = zeros.detach().requires_grad_(True)
        color, radii, depth, alpha = rasterize_views(t["means3D"], m2d, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.grad([color, depth], plist + [m2d], [gC, gD])
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True, group_by_stack_n=8).table(sort_by="self_cuda_time_total", row_limit=25,
                                                                               max_name_column_width=60, max_shapes_column_width=60, max_src_column_width=120))


if __name__ == "__main__":
    main()
