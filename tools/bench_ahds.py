#!/usr/bin/env python3
"""Full AHDS stage-1 training step on one GPU (BASELINE.json configs[2]): 4 views of 100k Gaussians at 1024^2 rendered
in one launch set -> bilinear 512^2 -> VAE encode (differentiable) -> ANPG: ControlNet(12) + U-Net(12) fp16 at 64^2
latents -> SDS loss + depth-sparsity loss -> backward through the VAE encoder and the rasterizer -> Adam on the six
Gaussian parameter groups.  Random-initialised SD1.5-shaped networks (no checkpoints in the build environment),
synthetic pose maps and prompt embeddings.  Prints one JSON line."""
import argparse
import json
import math
import os
import sys
import time
from argparse import ArgumentParser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def measure(steps=20, warmup=5, gaussians=100000, channels_last=True, flops=False, rank=0, world=1, device=None):
    """Runs the step `warmup + steps` times and returns the result dict.  With world > 1 (process group initialised by
    the caller) every rank trains on its own 4 cameras of the replicated Gaussians and the per-step exchange of
    parallel.exchange_step (gradients, densification statistics, depth maximum) runs inside the timed step."""
    import numpy as np
    import torch
    import scenes
    from scenes import orbit_c2w
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.guidance import GuidanceConfig, PromptEmbeddings, StableDiffusionGuidance
    from gaussianip_amd.renderer import render_views
    from gaussianip_amd.scene import Camera, GaussianModel
    from gaussianip_amd.utils import BasicPointCloud

    from gaussianip_amd import parallel
    dev = torch.device("cuda") if device is None else device
    torch.manual_seed(42)
    rng = np.random.default_rng(42)
    cam_rng = np.random.default_rng(42 + rank)     # replicated Gaussians, rank-specific cameras
    P, H, W, B = gaussians, 1024, 1024, 4
    gm = GaussianModel(0)
    pts = scenes.human_points(P, rng).astype(np.float32)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the model announces its size on stdout like the reference; keep stdout for the JSON line
        gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()))
    pipe = PipelineParams(ArgumentParser())
    bg = torch.zeros(3, device=dev)
    t0 = time.time()
    guidance = StableDiffusionGuidance(GuidanceConfig(channels_last=channels_last))
    g = torch.Generator(device=dev).manual_seed(1)
    tabs = [torch.randn(13, 77, 768, device=dev, generator=g) * 0.1 for _ in range(3)]
    prompts = PromptEmbeddings(*tabs, direction_fn=lambda el, az, c, v, d: ((az % 360) / 90).long())
    guidance.set_image_embeds(torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
                              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    setup_s = time.time() - t0
    pose = torch.rand(B, 512, 512, 3, device=dev, generator=g)

    def step(i):
        gm.update_learning_rate(i)
        rng = cam_rng
        el_h = rng.uniform(-30, 30, B).astype(np.float32)      # camera parameters live on the host (data module), like the reference
        az0 = rng.uniform(-180, 180)
        az_h = np.array([az0 + 90.0 * k for k in range(B)], np.float32)
        el = torch.from_numpy(el_h).to(dev, non_blocking=True)
        az = torch.from_numpy(az_h).to(dev, non_blocking=True)
        cams = [Camera(c2w=orbit_c2w(float(el_h[k]), float(az_h[k]), rng.uniform(1.3, 1.7)), data_device=dev, FoVy=math.radians(rng.uniform(40, 70)),
                       height=H, width=W) for k in range(B)]
        pkg = render_views(cams, gm, pipe, bg)
        rgb = pkg["render"].permute(0, 2, 3, 1)
        depth = pkg["depth_3dgs"].permute(0, 2, 3, 1)
        out = guidance(i, rgb, pose, prompts, True, torch.ones(B, device=dev), el, az, None, None)
        dmax = depth.detach().max()
        if world > 1:
            torch.distributed.all_reduce(dmax, op=torch.distributed.ReduceOp.MAX)   # GaussianIP.py:225: batch-global max
        opacity = depth / (dmax + 1e-5)
        loss = out["loss_sds"] + torch.sqrt(opacity ** 2 + 0.01).mean()
        gm.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        vs = pkg["viewspace_points"].grad.sum(0)
        radii = pkg["radii"].max(dim=0).values
        if world > 1:
            parallel.exchange_step([g_["params"][0] for g_ in gm.optimizer.param_groups], vs, radii, None, average=True)
        gm.max_radii2D = torch.max(gm.max_radii2D, radii.float())
        gm.add_densification_stats(vs, radii > 0)
        gm.optimizer.step()
        return loss

    for i in range(warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if world > 1:
        et = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(et, op=torch.distributed.ReduceOp.MAX)
        dt = float(et.item())

    # time the pieces (events on the current stream)
    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    lat = torch.randn(B, 4, 64, 64, device=dev)
    ctrl = pose.permute(0, 3, 1, 2)
    emb = torch.randn(3 * B, 81, 768, device=dev, dtype=torch.float16) * 0.1
    tt = torch.randint(20, 800, (B,), device=dev)
    den_ms = timed(lambda: guidance.forward_unet(torch.cat([lat] * 3), torch.cat([ctrl] * 3), torch.cat([tt] * 3), emb, True))
    img = torch.rand(B, 3, 512, 512, device=dev, requires_grad=True)

    def vae_fb():
        z = guidance.encode_images(img)
        z.sum().backward()
    vae_ms = timed(vae_fb)
    nflops = None
    if flops:
        from torch.utils.flop_counter import FlopCounterMode
        from gaussianip_amd.guidance import fused
        with fused.disabled(), FlopCounterMode(display=False) as fc:     # count on the plain-PyTorch path: the counter cannot see HIP launches
            guidance.forward_unet(torch.cat([lat] * 3), torch.cat([ctrl] * 3), torch.cat([tt] * 3), emb, True)
        nflops = fc.get_total_flops()
    flops = nflops
    out = {"metric": "ahds_train_steps_per_s", "value": round(world / dt, 3), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 2),
           "views_per_s": round(world * B / dt, 2), "n_gpus": world, "config": {"workload": "BASELINE.json configs[2]: 100k Gaussians, 1024^2, bs 4, SD1.5+ControlNet ANPG (batch 12, fp16), random-init weights", "gaussians": P},
           "denoise_ms": round(den_ms, 2), "vae_enc_fwd_bwd_ms": round(vae_ms, 2), "setup_s": round(setup_s, 1),
           "denoise_flops": flops, "denoise_tflops_per_s": None if not flops else round(flops / (den_ms * 1e-3) / 1e12, 1),
           "denoise_mfma_frac": None if not flops else round(flops / (den_ms * 1e-3) / 2.5e15, 4),   # fp16 dense peak ~2.5 PFLOP/s
           "data": "synthetic", "dtype": "f16 (networks) / f32 (raster)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=100000)
    ap.add_argument("--no-channels-last", action="store_true")
    ap.add_argument("--flops", action="store_true")
    args = ap.parse_args()
    print(json.dumps(measure(args.steps, args.warmup, args.gaussians, not args.no_channels_last, args.flops)), flush=True)


if __name__ == "__main__":
    main()
