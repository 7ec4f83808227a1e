#!/usr/bin/env python3
"""Stage-3 training step (GaussianIP.py:424-436): 4 of the 32 refine views rendered at 1024^2, cropped, halved, L1 + LPIPS
against the cached refined-image features, backward, Adam.  ms per step with the MFMA path; as a script also with plain ops.
`measure()` is the `config4.stage3` object of bench.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _setup(P):
    import contextlib
    from argparse import ArgumentParser
    import numpy as np
    import torch
    import scenes
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.scene.cameras import Camera
    from gaussianip_amd.system import create_refine_batch
    from gaussianip_amd.utils import BasicPointCloud
    dev = torch.device("cuda")
    pts = scenes.human_points(P, np.random.default_rng(0)).astype(np.float32)
    gm = GaussianModel(0)
    with contextlib.redirect_stdout(sys.stderr):
        gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    pipe = PipelineParams(ArgumentParser())
    bg = torch.ones(3, device=dev)
    b = create_refine_batch()
    cams = [Camera(c2w=b["c2w"][i], FoVy=float(b["fovy"][i]), height=1024, width=1024, data_device=dev) for i in range(32)]
    refined = torch.rand(32, 1024, 1024, 3, device=dev)
    return gm, pipe, bg, cams, refined


def _run(ctx, lp, n=10):
    import torch
    from gaussianip_amd.guidance.refine import VIEW_IDX_ALL
    from gaussianip_amd.system import StageThreeStep
    gm, pipe, bg, cams, refined = ctx
    st = StageThreeStep(gm, pipe, bg, cams, refined, VIEW_IDX_ALL, lambda_l1=10.0, lambda_lpips=15.0 if lp is not None else 0.0,
                        perceptual=lp, train_bs=4)

    def step():
        out = st.training_step()
        gm.optimizer.zero_grad(set_to_none=True)
        out["loss"].backward()
        gm.optimizer.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def measure(sizes=(100000, 1000000), steps=10):
    """{P: ms per stage-3 step (4 views at 1024^2, L1 + LPIPS-VGG fp16 on the MFMA convolutions, backward, Adam)}"""
    import torch
    from gaussianip_amd.guidance.perceptual import LPIPSVGG
    lp16 = LPIPSVGG().init_for_benchmark(0).prepare_inference(torch.device("cuda"))
    out = {"workload": "stage-3 step (GaussianIP.py:424-436): 4 of the 32 refine views at 1024^2, crop, half size, 10 L1 + 15 LPIPS-VGG (fp16, "
                       "cached target features), backward, Adam", "steps": steps}
    for P in sizes:
        ctx = _setup(P)
        out["ms_per_step_%dk_gaussians" % (P // 1000)] = {"l1_only": round(_run(ctx, None, steps), 3), "l1_plus_lpips": round(_run(ctx, lp16, steps), 3)}
        del ctx
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import torch
    from gaussianip_amd.guidance import fused
    from gaussianip_amd.guidance.perceptual import LPIPSVGG
    P = int(os.environ.get("P", 100000))
    ctx = _setup(P)
    dev = torch.device("cuda")
    print("L1 only                      : %.2f ms / step" % _run(ctx, None))
    lp16 = LPIPSVGG().init_for_benchmark(0).prepare_inference(dev)
    print("L1 + LPIPS fp16, MFMA convs  : %.2f ms / step" % _run(ctx, lp16))
    with fused.disabled():
        print("L1 + LPIPS fp16, MIOpen convs: %.2f ms / step" % _run(ctx, lp16))
    lp32 = LPIPSVGG().init_for_benchmark(0).to(dev)
    print("L1 + LPIPS fp32 (reference's precision), MIOpen: %.2f ms / step" % _run(ctx, lp32))
