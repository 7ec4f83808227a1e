#!/usr/bin/env python3
"""bench.py — the hot-path benchmark (contract: one JSON line from rank 0).

A "step" is one pass of the rasterizer hot path over one batch: V = 4 views of the SAME P = 100k Gaussians at
1024x1024, forward + backward (BASELINE.json configs[1]; batch of 4 cameras per step as configs/exp.yaml:59-61 and
threestudio/systems/GaussianIP.py:154-173), inputs resident in HBM, synthetic data (seed 42):
a 100k-point human-shaped surface (SMPL-X weights are licensed and absent), isotropic 3-NN scales, opacity 0.1, SH
degree 0 — the shipped init (gaussian_model.py:113-136) — and 4 cameras from the training ranges.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders its own 4 views of the replicated
Gaussian state (weak scaling, view-sharded data parallelism) and the per-step exchange of SURVEY.md §8e runs inside the
timed region over RCCL: all_reduce(sum) of the parameter gradients (one flat 14*P-float bucket), all_reduce(sum) of the
view-space gradient norms and all_reduce(max) of the radii.

"ahds": the full AHDS stage-1 training step of BASELINE.json configs[2] (render 4 views -> VAE encode -> ControlNet +
U-Net ANPG at batch 12 -> SDS -> backward -> Adam), measured by tools/bench_ahds.py after the raster timing; steps/s,
views/s and the denoise MFMA fraction.  Random-initialised SD1.5-shaped networks (no checkpoints without a network).

Extra objects: "roofline" (dominant kernel, live hipEvent durations on the launch stream) and "cpu_baseline" (the CPU
oracle = "port" of the reference algorithm, timed on this box's host cores on a bounded sample: 1 view fwd+bwd).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(P, K, R, T, N):
    """SURVEY.md §8d, per view: forward B_f and backward B_b."""
    b_f = P * (92 + 12 * K) + 88 * R + 8 * T + 24 * N
    b_b = 44 * R + 28 * N + P * (211 + 24 * K)
    return b_f, b_b


def stage_bytes(stage, P, K, R, T, N):
    """The same §8d terms split by kernel stage (per view); DESIGN.md §Kernels lists the derivation."""
    return {
        "preprocess": P * (44 + 12 * K) + 48 * P,
        "scan": 8 * T,
        "scatter": 12 * R,
        "tile_sort": 24 * R + 8 * R,
        "render_fwd": 44 * R + 24 * N,
        "render_bwd": 44 * R + 28 * N,
        "gather_bwd": P * (211 + 24 * K),
    }[stage]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=100000)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=10)
    ap.add_argument("--no-ahds", action="store_true", help="skip the full AHDS training-step measurement (configs[2])")
    ap.add_argument("--ahds-steps", type=int, default=10)
    ap.add_argument("--prewarm", type=int, default=60, help="untimed steps in front of the warmup steps (GPU clock ramp)")
    ap.add_argument("--no-trained", action="store_true", help="skip the secondary raster measurement on a trained-looking state")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import scenes
    from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
    from gaussianip_amd import rasterizer as R

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    local_rank %= max(torch.cuda.device_count(), 1)     # (a 2-rank functional check can share one GPU over gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GIP_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    P, H, W, V = args.gaussians, args.size, args.size, args.views
    K = 1  # SH coefficients per channel at the shipped sh_degree 0
    sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
    cams = scenes.train_cameras(V, seed=42 + rank, H=H, W=W)   # each rank (= view shard) gets its own cameras
    bg = torch.zeros(3, device=dev)
    sts = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
        sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
    gen = torch.Generator(device=dev).manual_seed(1234)
    gC = torch.randn((V, 3, H, W), device=dev, generator=gen) * 1e-3
    gD = torch.randn((V, 1, H, W), device=dev, generator=gen) * 1e-3
    names = ["means3D", "shs", "opacities", "scales", "rotations"]

    from gaussianip_amd import parallel
    plist = [t[n] for n in names]

    def step():
        m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        color, radii, depth, alpha = rasterize_views(t["means3D"], m2d, t["opacities"], sts, shs=t["shs"],
                                                     scales=t["scales"], rotations=t["rotations"])
        if world > 1:       # the MAX bucket holds forward outputs: its all-reduce overlaps the backward
            pending = parallel.exchange_forward_stats(radii, depth)
        grads = torch.autograd.grad([color, depth], plist + [m2d], [gC, gD])
        if world > 1:
            for p_, g_ in zip(plist, grads[:-1]):
                p_.grad = g_
            parallel.exchange_sum(plist, viewspace_grads=grads[-1])
            pending.wait()
        return color

    # Clock pre-warm (reported as config.prewarm_steps): a process starts on an idle GPU and the first ~30 ms of work run at
    # ramping clocks (measured: 0.67 -> 0.60 ms per synchronised step over the first 40 steps, tools/diag/step_settling.py);
    # with a short --warmup that ramp would sit inside the timed region.  These steps are the same step, untimed, in front
    # of the W warmup steps of the contract; the same count on every rank (the step has collectives at N > 1).
    for _ in range(args.prewarm):
        step()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        et = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())

    ms_per_step = elapsed / args.steps * 1e3
    views_total = world * V * args.steps
    mpix_s = views_total * H * W / elapsed / 1e6

    # forward only (SURVEY §8d asks for forward and forward+backward separately): the same 4-view launch set rendered
    # without a backward to follow.  Grad mode stays on so that the capacity check stays deferred as in training; an
    # output render under no_grad additionally waits for its header (one host round trip per call)
    def fwd_step():
        m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        return rasterize_views(t["means3D"], m2d, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    for _ in range(3):
        fwd_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd_step()
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t0) / args.steps * 1e3

    # ---- the same step on a TRAINED-looking state (secondary: the headline config is the init state SURVEY §8d prescribes)
    trained = None
    if rank == 0 and not args.no_trained:
        sct = scenes.trained_look(scenes.make_scene("human", P, seed=42, sh_degree=0), seed=7)
        tt_ = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sct.items()}
        pl_ = [tt_[n] for n in names]

        def step_t():
            m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
            color, radii, depth, alpha = rasterize_views(tt_["means3D"], m2d, tt_["opacities"], sts, shs=tt_["shs"],
                                                         scales=tt_["scales"], rotations=tt_["rotations"])
            torch.autograd.grad([color, depth], pl_ + [m2d], [gC, gD])
        for _ in range(max(args.warmup, 3)):
            step_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_t()
        torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0) / args.steps
        trained = {"ms_per_step": round(dtt * 1e3, 4), "mpix_per_s": round(V * H * W / dtt / 1e6, 1),
                   "state": "opacity 0.6, scales x U(1,3) per axis, random rotations / colours (tests/scenes.trained_look)"}
        del tt_, pl_

    # ---- BASELINE.json configs[3] layout of the same raster step (N > 1): the 4 views of ONE optimizer step sharded over
    # the ranks of a seed group (8 GPUs: 2 seed groups with their own process groups), gradients SUM-reduced inside it
    config3 = None
    if world > 1:
        shard = parallel.ViewSharding(V)
        cams3 = scenes.train_cameras(V, seed=42 + shard.seed_id, H=H, W=W)        # one camera set per seed group
        sts3 = [GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
            sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False)
            for i, c in enumerate(cams3) if i in shard.views]
        Vl = len(sts3)

        def step3():
            m2d = torch.zeros((Vl, P, 3), device=dev, requires_grad=True)
            color, radii, depth, alpha = rasterize_views(t["means3D"], m2d, t["opacities"], sts3, shs=t["shs"],
                                                         scales=t["scales"], rotations=t["rotations"])
            pending = parallel.exchange_forward_stats(radii, depth, group=shard.group)
            grads = torch.autograd.grad([color, depth], plist + [m2d], [gC[:Vl], gD[:Vl]])
            for p_, g_ in zip(plist, grads[:-1]):
                p_.grad = g_
            vs = grads[-1].sum(0)
            parallel.exchange_sum(plist, vs, group=shard.group)
            pending.wait()

        for _ in range(args.warmup):
            step3()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step3()
        dist.barrier()
        torch.cuda.synchronize()
        et = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        dt3 = float(et.item()) / args.steps
        config3 = {"layout": "4 views sharded over %d GPU(s) x %d seed group(s)" % (shard.group_size, shard.n_seed_groups),
                   "ms_per_step": round(dt3 * 1e3, 4), "optimizer_steps_per_s": round(shard.n_seed_groups / dt3, 2),
                   "views_per_s": round(shard.n_seed_groups * V / dt3, 2), "scaling": "strong within a seed group"}

    # ---- metric (i) of BASELINE.json: full AHDS training steps/s (configs[2]; at N>1 configs[3]'s view sharding) ----
    ahds = None
    if not args.no_ahds:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_ahds
        try:
            ahds = bench_ahds.measure(steps=args.ahds_steps, warmup=4, gaussians=P, flops=(rank == 0), rank=rank,
                                      world=world, device=dev)
        except Exception as e:  # the raster line above is the contract metric: never lose it to the secondary measurement
            ahds = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    out = None
    if rank == 0:
        # ---- roofline: live per-kernel durations (hipEvents on the launch stream) ----
        stages, num_rendered = R.profile_stages(
            t["means3D"].detach(), t["opacities"].detach(), sts, gC, gD, None, shs=t["shs"].detach(),
            scales=t["scales"].detach(), rotations=t["rotations"].detach(), iters=args.profile_iters)
        Rv = num_rendered / V
        T = ((H + 15) // 16) * ((W + 15) // 16)
        N = H * W
        b_f, b_b = algorithmic_bytes(P, K, Rv, T, N)
        kernel_stages = [s for s in stages if s != "clear"]
        dom = max(kernel_stages, key=lambda s: stages[s])
        dom_bytes = stage_bytes(dom, P, K, Rv, T, N) * V
        achieved = dom_bytes / (stages[dom] * 1e-3) / 1e9
        # HBM traffic and vector-issue counters come from committed rocprofv3 --pmc passes of THIS build (separate passes,
        # tools/collect_profiles.sh; FETCH_SIZE doubled and KiB units per the MI355X guide): they cannot be collected
        # inside this run, so the line names its source
        traffic, valu = None, None
        ppath = os.path.join(ROOT, "profiles", "pmc.json")
        if os.path.exists(ppath):
            try:
                pmc = json.load(open(ppath))
                c = pmc.get(dom, {})
                traffic = int(c["hbm_fetch_bytes"] + c["hbm_write_bytes"])
                # vector-issue roofline of the same kernel: a gfx950 SIMD retires one wave64 fp32 vector instruction
                # per 4 cycles (transcendental and packed forms take longer), the chip has 256 CUs x 4 SIMDs, the launch
                # lasted GRBM_GUI_ACTIVE / 8 XCD cycles: frac = 4 * SQ_INSTS_VALU / (1024 * cycles) is the share of the
                # launch's SIMD issue cycles that the kernel's vector instructions need at that best-case rate.
                cycles = c["GRBM_GUI_ACTIVE"] / 8.0
                valu = {"bound": "valu-issue", "kernel": "gip_%s_kernel" % dom, "wave_instructions_per_launch": int(c["SQ_INSTS_VALU"]),
                        "cycles_per_wave_instruction_assumed": 4,
                        "needed_simd_cycles": int(4.0 * c["SQ_INSTS_VALU"]), "available_simd_cycles": int(cycles * 1024),
                        "frac": round(4.0 * c["SQ_INSTS_VALU"] / (cycles * 1024), 4),
                        "busy_quad_cycles_counter": int(c["SQ_ACTIVE_INST_VALU"]), "launch_cycles": int(cycles),
                        "source": "profiles/pmc.json (rocprofv3 --pmc, this build)",
                        "formula": "4 * SQ_INSTS_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)"}
            except Exception:
                traffic, valu = None, None
        roofline = {"bound": "hbm", "kernel": "gip_%s_kernel" % dom, "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": "profiles/pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build)",
                    "algorithmic_bytes_per_launch": int(dom_bytes),
                    "avg_launch_ms": round(stages[dom], 4),
                    "stage_ms": {k: round(v, 4) for k, v in stages.items()},
                    "whole_step_GBs": round((b_f + b_b) * V / (ms_per_step * 1e-3) / 1e9, 2),
                    "whole_step_frac": round((b_f + b_b) * V / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "pair_evals_per_s_fwd": round(Rv * V * 256 / (stages["render_fwd"] * 1e-3), 0)}

        cpu = None
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            orc.build()
            ncores = os.cpu_count() or 1
            orc.set_threads(ncores)
            c0 = cams[0]
            ro = orc.RasterOracle()
            kw = dict(image_height=H, image_width=W, tanfovx=c0["tanfovx"], tanfovy=c0["tanfovy"],
                      bg=np.zeros(3, np.float32), scale_modifier=1.0, viewmatrix=c0["viewmatrix"],
                      projmatrix=c0["projmatrix"], sh_degree=0, campos=c0["campos"], means3D=sc["means3D"],
                      opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
            gc_h, gd_h = gC[0].cpu().numpy(), gD[0].cpu().numpy()
            ro.forward(**kw)  # warm-up (page-in)
            ro.backward(gc_h, gd_h, None)
            t_fwd, t_both = [], []
            for _ in range(5):                                   # median of 5 after one warm-up
                c1 = time.perf_counter()
                ro.forward(**kw)
                c2 = time.perf_counter()
                ro.backward(gc_h, gd_h, None)
                c3 = time.perf_counter()
                t_fwd.append(c2 - c1)
                t_both.append(c3 - c1)
            t_fwd.sort()
            t_both.sort()
            reps, tt = 1, t_both[2]
            model = ""
            try:
                with open("/proc/cpuinfo") as f:
                    model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
            except OSError:
                pass
            cpu = {"value": round(reps * H * W / tt / 1e6, 3), "unit": "Mpix/s", "cores": ncores, "cpu_model": model, "kind": "port",
                   "forward_only_value": round(H * W / t_fwd[2] / 1e6, 3),
                   "sample": "median of 5 x (1 view forward, then backward; P=%d, %dx%d) after one warm-up, on the C oracle "
                             "(oracle/raster_oracle.c, OpenMP over tiles, %d threads); value = forward+backward" % (P, H, W, ncores)}

        out = {"metric": "raster_fwd_bwd_mpix_per_s", "value": round(mpix_s, 2), "unit": "Mpix/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[1]: %d Gaussians (synthetic human surface, SMPL-X-style "
                                      "init), %dx%d, %d views/step per GPU, raster forward+backward" % (P, H, W, V),
                          "gaussians": P, "height": H, "width": W, "views_per_step_per_gpu": V, "prewarm_steps": args.prewarm,
                          "num_rendered_per_view": int(Rv), "sh_degree": 0,
                          "parallelism": "view-sharded dp%d" % world},
               "raster_steps_per_s": round(1e3 / ms_per_step, 3), "views_per_s": round(views_total / elapsed, 2),
               "forward_only": {"ms_per_step": round(fwd_ms, 4), "mpix_per_s_per_gpu": round(V * H * W / fwd_ms / 1e3, 1)},
               "trained_state": trained,
               "config3_layout": config3,
               "roofline": roofline, "roofline_valu": valu, "cpu_baseline": cpu}
    if rank == 0:
        out["ahds"] = ahds
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
