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


_guidance_cache = {}     # (channels_last, device) -> StableDiffusionGuidance: the measurements of one process share ONE set of networks
                         # (deterministic seeds: a second instance would hold identical weights; building it costs ~24 s)


def cached_guidance():
    """The guidance object the measurements of this process have built (None before the first one): bench.py hands it to the
    configs[4] refine measurement."""
    return next(iter(_guidance_cache.values()), None)


def measure(steps=20, warmup=5, gaussians=100000, channels_last=True, flops=False, rank=0, world=1, device=None, amp=True,
            fused_adam=True, layout="views", trained=False, proxy_group=0, pieces=True, group_size=None):
    """Runs the step `warmup + steps` times and returns the result dict.  The timed step is the reference's
    training_step + optimizer step as the system runs them (system.StageOneStep.training_step / optimizer_step):
    learning-rate update, render of the 4 cameras, OpenPose pose maps drawn on the GPU from the batch's mvp matrices,
    view-dependent prompt lookup, guidance call, loss assembly, backward, densification statistics, Adam — with a
    GradScaler when `amp` (the reference trains with `precision: 16-mixed`, configs/exp.yaml:193; its scaler.step() is
    the one host synchronisation of the step, as in the reference).
    `trained`: the Gaussians get the trained-looking state of tests/scenes.trained_look (opacity 0.6, anisotropic 1-3x scales,
    random rotations / colours) instead of the init state.  `proxy_group` = k (1 GPU): this process runs the shard of ONE
    rank of a k-rank seed group of configs[3] (4 / k views, batch 3 x 4 / k denoise) with no collectives.  `pieces`: also
    time the denoise and the VAE on their own.
    With world > 1 (process group initialised by the caller) and layout "views" — BASELINE.json configs[3] — the 4 views
    of an optimizer step are sharded over the ranks of a seed group (2 GPUs: 2 views each; 4: 1 view each; 8: two
    independent seed groups of 4, each with its own process group: parallel.ViewSharding), the guidance runs on the local
    views (batch 3 x local views) and the per-step exchange runs inside the timed step.  Layout "replicas": every rank
    trains on its own 4 cameras (a 4 x world batch) with averaged gradients."""
    import numpy as np
    import torch
    import scenes
    if os.environ.get("GIP_TUNABLEOP_FILE"):
        # (re-)tuning run: PyTorch TunableOp picks the hipBLASLt / rocBLAS solution per GEMM shape and writes $GIP_TUNABLEOP_FILE
        # (GIP_TUNABLEOP_TUNE=1 tunes the shapes the file does not hold — during the eager first call of every shape, before any
        # capture).  How gaussianip_amd/guidance/tunableop_gfx950.csv was made (then filtered to hipBLASLt solutions of half GEMMs).
        import torch.cuda.tunable as tun
        tun.enable(True)
        tun.set_filename(os.environ["GIP_TUNABLEOP_FILE"])
        tun.tuning_enable(os.environ.get("GIP_TUNABLEOP_TUNE", "0") == "1")
        if os.path.exists(os.environ["GIP_TUNABLEOP_FILE"]):
            tun.read_file(os.environ["GIP_TUNABLEOP_FILE"])
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    from gaussianip_amd.guidance.prompts import PromptProcessor
    from gaussianip_amd.poser import Skeleton
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.system import StageOneStep
    from gaussianip_amd.utils import BasicPointCloud

    from gaussianip_amd import parallel
    dev = torch.device("cuda") if device is None else device
    torch.manual_seed(42)
    rng = np.random.default_rng(42)
    shard = parallel.ViewSharding(4, group_size=group_size) if (world > 1 and layout == "views") else None
    if proxy_group:
        shard = parallel.ViewSharding(4, rank=0, world=proxy_group, make_groups=False)       # inactive: no process group -> no collectives
    # views layout: every rank of a seed group draws the SAME 4 cameras (seed offset per seed group, launch.py:80) and
    # renders its share; replicas layout: rank-specific cameras
    cam_rng = np.random.default_rng(42 + (shard.seed_id if shard is not None else rank))
    P, H, W, B = gaussians, 1024, 1024, 4
    gm = GaussianModel(0)
    pts = scenes.human_points(P, rng).astype(np.float32)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the model announces its size on stdout like the reference; keep stdout for the JSON line
        gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
    if trained:
        tl = scenes.trained_look(dict(means3D=pts, opacities=np.full((P, 1), 0.1, np.float32), shs=np.zeros((P, 1, 3), np.float32),
                                      scales=np.exp(gm._scaling.detach().cpu().numpy()), rotations=np.tile(np.float32([1, 0, 0, 0]), (P, 1))), seed=7)
        with torch.no_grad():
            gm._opacity.copy_(torch.logit(torch.from_numpy(tl["opacities"]).clamp(1e-4, 1 - 1e-4)).to(gm._opacity.device))
            gm._scaling.copy_(torch.log(torch.from_numpy(tl["scales"])).to(gm._scaling.device))
            gm._rotation.copy_(torch.from_numpy(tl["rotations"]).to(gm._rotation.device))
            gm._features_dc.copy_(torch.from_numpy(tl["shs"]).reshape(gm._features_dc.shape).to(gm._features_dc.device))
    gm.training_setup(OptimizationParams(ArgumentParser()), fused=fused_adam)
    skel = Skeleton(dev)
    skel.scale(-10)
    stage = StageOneStep(gm, PipelineParams(ArgumentParser()), torch.zeros(3, device=dev), skeleton=skel)
    t0 = time.time()
    g = torch.Generator(device=dev).manual_seed(1)
    tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    gkey = (bool(channels_last), str(dev))
    guidance = _guidance_cache.get(gkey)
    if guidance is None:
        guidance = _guidance_cache[gkey] = StableDiffusionGuidance(GuidanceConfig(channels_last=channels_last), image_embeds_provider=lambda gd: tokens)
    pp = PromptProcessor("a person wearing a coat", lambda texts: torch.randn(len(texts), 77, 768, device=dev, generator=g).half() * 0.1,
                         negative_prompt="blurry")
    guidance.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)
    prompt_utils = pp()
    setup_s = time.time() - t0
    scaler = torch.amp.GradScaler("cuda") if amp else None
    if shard is not None:
        stage.sharding = shard
    elif world > 1:
        def all_max(x):
            torch.distributed.all_reduce(x, op=torch.distributed.ReduceOp.MAX)
            return x
        stage.depth_max_reduce = all_max
    pose = torch.rand(B, 512, 512, 3, device=dev, generator=g)

    def exchange(st):
        # replicated Gaussians, rank-specific cameras = one global batch of 4 * world views: gradients averaged, the
        # view-space gradient vectors summed, radii maximum (GaussianIP.py:452-457 over all ranks' views)
        vs = st.viewspace_points.grad.sum(0)
        parallel.exchange_step([g_["params"][0] for g_ in gm.optimizer.param_groups], vs, st.radii, None, average=True)
        st.viewspace_grad_sum = vs
        st.visibility_filter = st.visibility(st.radii)

    def step(i):
        # the data module's batch: CPU tensors (camera matrices AND the per-view scalars); what the GPU needs is uploaded by its consumer
        batch = scenes.train_batch(cam_rng, B, H, W, device=None if os.environ.get("GIP_HOST_BATCH", "1") == "1" else dev)
        loss, out, gout = stage.training_step(i, batch, guidance, prompt_utils, True)
        stage.optimizer_step(loss, i, scaler=scaler, exchange=(shard.exchange if shard is not None else exchange) if (world > 1 or proxy_group) else None)
        step.last_loss = loss.detach()
        return loss

    for i in range(warmup):
        step(i)
    # GradScaler bookkeeping: a step whose scaled gradients overflow is SKIPPED by scaler.step() and halves
    # the scale.  The default initial scale (65536, what Lightning's 16-mixed uses) overflows the fp16 VAE backward of these
    # random-initialised networks, so the first steps after start-up are skipped steps: warm up until the scale has settled,
    # then count the skipped steps INSIDE the timed region from the scale before / after it (growth interval 2000 steps)
    extra_warm = 0
    scale_trace = []
    if scaler is not None:
        prev = scaler.get_scale()
        scale_trace.append(prev)
        while extra_warm < 24:
            step(warmup + extra_warm)
            extra_warm += 1
            cur = scaler.get_scale()
            scale_trace.append(cur)
            if cur == prev:
                break
            prev = cur
    warmup += extra_warm
    scale_before = scaler.get_scale() if scaler is not None else None
    timeline = {}
    if os.environ.get("GIP_HOST_TIMELINE"):
        # where does the HOST spend the step?  perf_counter / thread_time around the phase boundaries, no synchronisation added:
        # a phase whose wall time is far above its CPU time is where the host BLOCKS on the GPU
        def wrap(obj, name, label):
            fn = getattr(obj, name)

            def w(*a, **k):
                t_, c_ = time.perf_counter(), time.thread_time()
                try:
                    return fn(*a, **k)
                finally:
                    d = timeline.setdefault(label, [0.0, 0.0, 0])
                    d[0] += time.perf_counter() - t_
                    d[1] += time.thread_time() - c_
                    d[2] += 1
            setattr(obj, name, w)
        wrap(scenes, "train_batch", "0 data batch (host)")
        wrap(stage, "forward", "1 render + pose maps")
        wrap(skel, "openpose_draw", "1a  pose maps")
        wrap(guidance, "encode_images", "2a  VAE encode")
        wrap(guidance, "forward_unet", "2b  ControlNet + U-Net")
        wrap(type(guidance), "__call__", "2 guidance (whole)")
        wrap(torch.Tensor, "backward", "3 backward")
        if scaler is not None:
            wrap(scaler, "unscale_", "4a  unscale")
            wrap(scaler, "step", "4b  Adam")
            wrap(scaler, "update", "4c  scaler update")
        wrap(stage, "optimizer_step", "4 optimizer step (whole, incl. backward)")
    if os.environ.get("GIP_TORCH_PROFILE"):
        # op-level table of two steady-state steps (which aten op, with which shapes, from which line launched the glue kernels)
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            step(warmup)
            step(warmup + 1)
            torch.cuda.synchronize()
        with open(os.environ["GIP_TORCH_PROFILE"], "w") as f:
            f.write(prof.key_averages(group_by_input_shape=True, group_by_stack_n=6).table(
                sort_by="self_cuda_time_total", row_limit=120, max_name_column_width=50, max_shapes_column_width=70, max_src_column_width=110))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_s = host_cpu_s = 0.0
    for i in range(steps):
        h0, c0 = time.perf_counter(), time.thread_time()
        step(warmup + i)
        host_s += time.perf_counter() - h0          # wall time of the enqueue calls of a step (no synchronisation inside, but the
        host_cpu_s += time.thread_time() - c0       # runtime blocks the host once it is a queue's depth ahead); CPU time of the thread
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    last_loss = float(step.last_loss) if getattr(step, "last_loss", None) is not None else None
    amp_info = None
    if scaler is not None:
        scale_after = scaler.get_scale()
        amp_info = {"init_scale": scale_trace[0] * (2 ** 0), "scale_at_timed_region": scale_before, "scale_after": scale_after,
                    "skipped_steps_in_timed_region": (0 if scale_after >= scale_before else int(round(math.log2(scale_before / scale_after)))),
                    "settling_warmup_steps": extra_warm, "scale_trace_during_warmup": scale_trace}
    host_ms = host_s / steps * 1e3
    for label in sorted(timeline):
        d = timeline[label]
        print("host timeline  %-44s wall %7.2f ms  cpu %6.2f ms  (%d calls / step)" % (label, d[0] / steps * 1e3, d[1] / steps * 1e3, d[2] // steps), file=sys.stderr)
    host_cpu_ms = host_cpu_s / steps * 1e3
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

    if not pieces:
        return {"value": round((shard.n_seed_groups if shard is not None and not proxy_group else 1) / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                "host_enqueue_ms_per_step": round(host_ms, 2), "host_cpu_ms_per_step": round(host_cpu_ms, 2),
                "grad_scaler": amp_info, "final_loss": last_loss}
    lat = torch.randn(B, 4, 64, 64, device=dev)
    ctrl = pose.permute(0, 3, 1, 2)
    emb = torch.randn(3 * B, 81, 768, device=dev, dtype=torch.float16) * 0.1
    tt = torch.randint(20, 800, (B,), device=dev)

    # the call the STEP makes (ipa_guidance._call_fused): batch 3 B with replicas = 3 — the layers in front of the first
    # cross-attention run on one copy of the three ANPG branches.  Timed and FLOP-counted on this same call (VERDICT r4 weak 12:
    # round 4 timed the default replicas = 1, a path the step does not take).
    def denoise():
        return guidance.forward_unet(torch.cat([lat] * 3), ctrl, torch.cat([tt] * 3), emb, True, replicas=3)
    den_ms = timed(denoise)
    img = torch.rand(B, 3, 512, 512, device=dev, requires_grad=True)

    def vae_fb():
        z = guidance.encode_images(img)
        z.sum().backward()
    vae_ms = timed(vae_fb)
    nflops = vflops = None
    if flops:
        from torch.utils.flop_counter import FlopCounterMode
        from gaussianip_amd.guidance import fused
        with fused.disabled(), FlopCounterMode(display=False) as fc:     # count on the plain-PyTorch path: the counter cannot see HIP launches
            denoise()
        nflops = fc.get_total_flops()
        # "+ VAE encoder fwd x4 and bwd x4".  The weights are frozen, so the backward is data gradients only; counted on
        # the plain-PyTorch path like the denoise (forward + backward of one encode_images call on the B images)
        img.grad = None
        with fused.disabled(), FlopCounterMode(display=False) as fv:
            vae_fb()
        vflops = fv.get_total_flops()
        img.grad = None
    flops = nflops
    if shard is not None:
        views_per_step, opt_steps = B, shard.n_seed_groups          # every seed group takes one optimizer step per dt
        lay = "configs[3]: 4 views sharded over %d GPU(s) x %d independent seed group(s)" % (shard.group_size, shard.n_seed_groups)
    else:
        views_per_step, opt_steps = world * B, 1
        lay = "single GPU" if world == 1 else "replicas: %d ranks x 4 cameras, one averaged optimizer step" % world
    out = {"metric": "ahds_train_steps_per_s", "value": round(opt_steps / dt, 3), "unit": "optimizer steps/s", "ms_per_step": round(dt * 1e3, 2),
           "views_per_s": round(opt_steps * views_per_step / dt, 2), "views_per_optimizer_step": views_per_step, "n_gpus": world,
           "layout": lay, "amp_gradscaler": bool(amp), "grad_scaler": amp_info, "final_loss": last_loss, "fused_adam": bool(fused_adam),
           "timed_step": "lr update + render 4 views + GPU pose maps + prompt lookup + VAE/ControlNet/U-Net ANPG + loss + backward + densification stats + Adam", "config": {"workload": "BASELINE.json configs[2]: 100k Gaussians, 1024^2, bs 4, SD1.5+ControlNet ANPG (batch 12, fp16), random-init weights", "gaussians": P},
           "host_enqueue_ms_per_step": round(host_ms, 2), "host_cpu_ms_per_step": round(host_cpu_ms, 2), "denoise_ms": round(den_ms, 2), "vae_enc_fwd_bwd_ms": round(vae_ms, 2), "setup_s": round(setup_s, 1),
           "denoise_call": "forward_unet(batch %d, replicas=3): the call of the step" % (3 * B),
           "denoise_flops": flops, "denoise_tflops_per_s": None if not flops else round(flops / (den_ms * 1e-3) / 1e12, 1),
           "denoise_mfma_frac": None if not flops else round(flops / (den_ms * 1e-3) / 2.5e15, 4),   # fp16 dense peak ~2.5 PFLOP/s
           "vae_flops": vflops, "vae_tflops_per_s": None if not vflops else round(vflops / (vae_ms * 1e-3) / 1e12, 1),
           "vae_mfma_frac": None if not vflops else round(vflops / (vae_ms * 1e-3) / 2.5e15, 4),
           "networks_mfma_frac": None if not (flops and vflops) else round((flops + vflops) / ((den_ms + vae_ms) * 1e-3) / 2.5e15, 4),
           "pieces_note": "denoise_ms and vae_enc_fwd_bwd_ms are timed on their own, back to back (no other work between the calls): "
                          "their sum need not equal the step they are pieces of",
           "data": "synthetic", "dtype": "f16 (networks) / f32 (raster)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=100000)
    ap.add_argument("--no-channels-last", action="store_true")
    ap.add_argument("--flops", action="store_true")
    ap.add_argument("--no-amp", action="store_true", help="no GradScaler (the reference trains with 16-mixed)")
    ap.add_argument("--proxy-group", type=int, default=0, help="run ONE rank's shard of a k-rank configs[3] seed group on this GPU (no collectives)")
    ap.add_argument("--trained", action="store_true", help="trained-looking Gaussian state instead of the init state")
    ap.add_argument("--plain-adam", action="store_true", help="torch's default Adam like the reference (GradScaler.step then synchronises)")
    args = ap.parse_args()
    print(json.dumps(measure(args.steps, args.warmup, args.gaussians, not args.no_channels_last, args.flops, amp=not args.no_amp, fused_adam=not args.plain_adam,
                             proxy_group=args.proxy_group, trained=args.trained)), flush=True)


if __name__ == "__main__":
    main()
