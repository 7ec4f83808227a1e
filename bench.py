#!/usr/bin/env python3
"""bench.py — the hot-path benchmark (contract: one JSON line from rank 0).

A "step" is one pass of the rasterizer hot path over one batch: V = 4 views of the SAME P = 100k Gaussians at
1024x1024, forward + backward (BASELINE.json configs[1]; batch of 4 cameras per step as configs/exp.yaml:59-61 and
threestudio/systems/GaussianIP.py:154-173), inputs resident in HBM, synthetic data (seed 42):
a 100k-point human-shaped surface (SMPL-X weights are licensed and absent), isotropic 3-NN scales, opacity 0.1, SH
degree 0 — the shipped init (gaussian_model.py:113-136) — and 4 cameras from the training ranges.

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[3] — the 4 views of ONE optimizer
step are sharded over the ranks of a seed group (2 GPUs: 2 views each, 4 GPUs: 1 view each, 8 GPUs: 4 views x 2 seed
groups; parallel.ViewSharding) and the per-step exchange of SURVEY.md §8e runs inside the timed region over RCCL, inside
each seed group: all_reduce(sum) of the parameter gradients + view-space gradient norms (one flat bucket) and
all_reduce(max) of the radii / depth maximum.  `value` = all ranks' views x H x W per second ("scaling": "strong": the
work of an optimizer step is fixed, a rank's share shrinks with N).  The round-1/2 layout (every rank its own 4 cameras,
a 4 x N batch) is reported beside it as `replicas_layout`.

Timing: W warmup steps, then R = --repeats windows of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides and MAX-reduced over the ranks; `ms_per_step` / `value` are the MEDIAN window (config.repeats, all windows in
`window_ms_per_step`): 20 steps are a 10 ms region, a single one is a +-2 % instrument.

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


def raster_source_hash():
    """sha256[:16] over the rasterizer's kernel sources, its internal header, the C-ABI header and the Makefile (flags):
    what profiles/pmc.json is stamped with.  Identical sources give identical kernels whichever box built them."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "gaussianip_amd", "csrc")
    for f in ("preprocess.hip", "binning.hip", "api.hip", "render_forward.hip", "render_backward.hip", "gather_backward.hip", "sh_mfma.hip",
              "gip_internal.h", os.path.join("..", "..", "include", "gip_raster.h")):
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(csrc, "Makefile")) as fh:      # the flag lines the raster objects are compiled with (libgip_nn's rules do not count)
        for ln in fh:
            if ln.split("=")[0].strip().split(" ")[0] in ("ARCH", "COMMON", "EXACT", "FAST", "SRCS_EXACT", "SRCS_FAST") or ln.startswith("NOSLP_"):
                h.update(ln.strip().encode())
    return h.hexdigest()[:16]


def physical_cores():
    """(physical cores, hardware threads) of this host: distinct (physical id, core id) pairs of /proc/cpuinfo."""
    threads = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        if pairs:
            return min(len(pairs), threads), threads
    except OSError:
        pass
    return threads, threads


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


def measure_config4(dev, P=1000000, H=1024, W=1024, V=12, iters=5):
    """BASELINE.json configs[4], render part, where the HBM roofline is meaningful: 1M Gaussians (the 100k init split / cloned up
    to 1M: scales / 1.6 as gaussian_model.py:371, opacity 0.6), 1024^2, ONE 12-view launch set of the 36-view orbit (elevation 5,
    distance 1.8, fovy 70; configs/exp.yaml:37-40).  Per stage: live hipEvent duration on the launch stream, algorithmic bytes
    (SURVEY §8d terms split by stage x V) and their fraction of 8 TB/s; forward and forward+backward wall time of the set."""
    import numpy as np
    import torch
    import scenes
    from gaussianip_amd import GaussianRasterizationSettings, rasterize_views
    from gaussianip_amd import rasterizer as R
    sc = scenes.make_scene("human", P, seed=42)
    sc["scales"] = (sc["scales"] / 1.6).astype(np.float32)
    sc["opacities"][:] = 0.6
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev).contiguous() for k, v in sc.items()}
    bg = torch.zeros(3, device=dev)
    cams = [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(V)]
    sts = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
        sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cams]
    gen = torch.Generator(device=dev).manual_seed(4321)
    gC = torch.randn((V, 3, H, W), device=dev, generator=gen) * 1e-3
    gD = torch.randn((V, 1, H, W), device=dev, generator=gen) * 1e-3
    names = ["means3D", "shs", "opacities", "scales", "rotations"]
    tg = {k: v.clone().requires_grad_(True) for k, v in t.items()}

    def fwd():
        with torch.no_grad():
            return rasterize_views(t["means3D"], None, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])

    def fwd_bwd():
        color, radii, depth, alpha = rasterize_views(tg["means3D"], None, tg["opacities"], sts, shs=tg["shs"], scales=tg["scales"],
                                                     rotations=tg["rotations"])
        torch.autograd.grad([color, depth], [tg[n] for n in names], [gC, gD])

    def wall(fn, n=5):
        fn()
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    f_ms, fb_ms = wall(fwd), wall(fwd_bwd)
    stages, num_rendered = R.profile_stages(t["means3D"], t["opacities"], sts, gC, gD, None, shs=t["shs"], scales=t["scales"],
                                            rotations=t["rotations"], iters=iters)
    Rv = num_rendered / V
    T, N, K = ((H + 15) // 16) * ((W + 15) // 16), H * W, 1
    b_f, b_b = algorithmic_bytes(P, K, Rv, T, N)
    per_stage = {}
    for st_, ms in stages.items():
        if st_ == "clear":
            per_stage[st_] = {"ms": round(ms, 4)}
            continue
        by = stage_bytes(st_, P, K, Rv, T, N) * V
        per_stage[st_] = {"ms": round(ms, 4), "algorithmic_bytes": int(by), "GBs": round(by / (ms * 1e-3) / 1e9, 1),
                          "frac_of_8TBs": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    fwd_stage_ms = sum(stages[k] for k in ("clear", "preprocess", "scan", "scatter", "tile_sort", "render_fwd"))
    all_stage_ms = sum(stages.values())
    return {"workload": "BASELINE.json configs[4], render part: %d Gaussians (post-densify look), %dx%d, one %d-view launch set of the 36-view orbit" % (P, H, W, V),
            "gaussians": P, "views_per_launch_set": V, "num_rendered_per_view": int(Rv),
            "forward_ms_per_set": round(f_ms, 3), "forward_backward_ms_per_set": round(fb_ms, 3),
            "forward_views_per_s": round(V / f_ms * 1e3, 1), "forward_backward_views_per_s": round(V / fb_ms * 1e3, 1),
            "stages_instrumented": per_stage,
            "stage_note": "hipEvent pairs around every stage on the launch stream (gip_raster_*_profiled): the events add ~3 % over the un-instrumented set",
            "algorithmic_bytes_per_view": {"forward": int(b_f), "backward": int(b_b)},
            "forward_frac_of_8TBs": round(b_f * V / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "whole_step_frac_of_8TBs": round((b_f + b_b) * V / (fb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "sum_stage_ms": {"forward": round(fwd_stage_ms, 4), "forward_backward": round(all_stage_ms, 4)}}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _visible_gpus():
    """Device count, asked of a CHILD process: the launcher itself must never initialise the GPU runtime (it only starts
    rank processes and relays their output)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:      # noqa: BLE001
        return 0


def spawn_ranks(n, cmd, env=None, out=None, err=None, timeout=None):
    """Start `n` fresh rank processes of `cmd` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their
    environment, the contract torch.distributed.run gives its workers), wait for all of them, and relay ONE line to `out`:
    the last line of rank 0's stdout that is a JSON object with a "metric" key.  Everything else the ranks print on stdout
    (library banners such as "[Gloo] Rank ..." included) goes to `err`, prefixed by the rank.  Returns the exit code: 0 only
    if every rank exited 0 and rank 0 produced its line; the first failing rank's code otherwise (the remaining ranks are
    terminated by PID, and killed if they ignore that for 10 s).  `timeout` (seconds, None = none): overall deadline — when
    it passes with ranks still running (all of them blocked in a rendezvous, say), they are stopped the same way, their
    output is dumped and the code is 124.  The calling process never touches the GPU."""
    import subprocess
    import threading
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(_free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["WORLD_SIZE"] = str(n)
    procs, lines = [], [[] for _ in range(n)]

    def pump(rank, pipe):
        for ln in pipe:
            lines[rank].append(ln.rstrip("\n"))
    threads = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0")
        p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, text=True, bufsize=1)
        t = threading.Thread(target=pump, args=(r, p.stdout), daemon=True)
        t.start()
        procs.append(p)
        threads.append(t)
    rc, pending = 0, set(range(n))
    deadline = None if not timeout else time.monotonic() + float(timeout)
    kill_at = None                      # ranks that were sent SIGTERM and are still alive at this time get SIGKILL

    def stop_others(reason):
        nonlocal kill_at
        print("bench.py launcher: %s; stopping the other ranks" % reason, file=err)
        for q in pending:
            procs[q].terminate()          # exact PIDs this launcher started
        kill_at = time.monotonic() + 10.0

    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                stop_others("rank %d exited with code %d" % (r, code))
        if pending and deadline is not None and time.monotonic() > deadline and rc == 0:
            # every rank may be blocked (a rendezvous that never completes, a peer stuck in a kernel): no rank exits, so
            # nothing above ever fires — the launcher gives up itself, dumps what the ranks printed and fails
            rc = 124
            deadline = None
            stop_others("no result after %.0f s (--launch-timeout / GIP_BENCH_LAUNCH_TIMEOUT)" % float(timeout))
        if pending and kill_at is not None and time.monotonic() > kill_at:
            for q in pending:             # SIGTERM ignored (blocked inside the driver): escalate
                procs[q].kill()
            kill_at = None
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    line = None
    for ln in lines[0]:
        if ln.startswith("{"):
            try:
                if "metric" in json.loads(ln):
                    line = ln
                    continue
            except ValueError:
                pass
        print("[rank 0] %s" % ln, file=err)
    for r in range(1, n):
        for ln in lines[r]:
            print("[rank %d] %s" % (r, ln), file=err)
    if rc == 0 and line is None:
        print("bench.py launcher: rank 0 printed no result line", file=err)
        rc = 1
    if rc == 0:
        print(line, file=out, flush=True)
    return rc


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
    ap.add_argument("--prewarm", type=int, default=0, help="extra untimed steps in front of the warmup steps (reported)")
    ap.add_argument("--repeats", type=int, default=11, help="timed windows of --steps steps each; the median is reported")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact_lists mode measurement")
    ap.add_argument("--no-proxy", action="store_true", help="skip the 1-GPU proxy of one configs[3] rank's shard")
    ap.add_argument("--no-trained", action="store_true", help="skip the secondary raster measurement on a trained-looking state")
    ap.add_argument("--no-config4", action="store_true", help="skip the 1M-Gaussian per-stage roofline object (BASELINE configs[4])")
    ap.add_argument("--no-refine", action="store_true", help="skip configs[4]'s refine-pass / stage-3 measurement (config4.refine, config4.stage3)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("GIP_BENCH_LAUNCH_TIMEOUT", "3600")),
                    help="--gpus N launcher: seconds after which still-running ranks are stopped and the run fails (0 = never)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it, outside torch.distributed.run: THIS process becomes the
        # launcher — it starts N fresh rank processes of this file (one per GPU, RCCL), never initialises the GPU itself
        # (no re-exec of a process that has touched the GPU), relays rank 0's one JSON line and fails if any rank fails
        env = dict(os.environ)
        ndev = _visible_gpus()
        env["GIP_BENCH_VISIBLE_GPUS"] = str(ndev)
        if ndev < args.gpus and "GIP_DIST_BACKEND" not in env:
            # fewer GPUs than ranks (a functional check on a 1-GPU box): RCCL cannot put two ranks on one device, gloo can;
            # the line says so (config.gpus_visible / config.backend) — it is not an N-GPU measurement
            env["GIP_DIST_BACKEND"] = "gloo"
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env, timeout=args.launch_timeout or None))

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
    backend = None
    result_out = sys.stdout
    if world > 1:
        # stdout carries the ONE result line and nothing else: communication libraries announce themselves on file descriptor 1
        # ("[Gloo] Rank 0 is connected to 1 peer ranks"), so from here on descriptor 1 is stderr and the line goes to a duplicate
        # of the original stdout
        sys.stdout.flush()
        result_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GIP_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    P, H, W, V = args.gaussians, args.size, args.size, args.views
    K = 1  # SH coefficients per channel at the shipped sh_degree 0
    sc = scenes.make_scene("human", P, seed=42, sh_degree=0)
    from gaussianip_amd import parallel
    bg = torch.zeros(3, device=dev)

    def settings(cam_list):
        return [GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=torch.from_numpy(c["viewmatrix"]).to(dev), projmatrix=torch.from_numpy(c["projmatrix"]).to(dev),
            sh_degree=0, campos=torch.from_numpy(c["campos"]).to(dev), prefiltered=False, debug=False) for c in cam_list]

    shard = parallel.ViewSharding(V) if world > 1 else None
    # N = 1: the 4 cameras of the step.  N > 1 (configs[3]): every rank of a seed group draws the SAME 4 cameras (seed offset
    # per seed group, launch.py:80) and renders its share of them
    cams = scenes.train_cameras(V, seed=42 + (shard.seed_id if shard is not None else 0), H=H, W=W)
    sts = settings(cams if shard is None else [c for i, c in enumerate(cams) if i in shard.views])
    Vl = len(sts)                                   # views this rank renders per step
    # inputs resident in HBM in the layout the interface takes (contiguous float32, as GaussianModel's getters return them): the
    # scene generator's isotropic scales are a numpy broadcast, which the wrapper would otherwise re-pack on every call
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev).contiguous().requires_grad_(True) for k, v in sc.items()}
    gen = torch.Generator(device=dev).manual_seed(1234)
    gC = torch.randn((V, 3, H, W), device=dev, generator=gen) * 1e-3
    gD = torch.randn((V, 1, H, W), device=dev, generator=gen) * 1e-3
    names = ["means3D", "shs", "opacities", "scales", "rotations"]
    plist = [t[n] for n in names]

    def make_step(st_list, group, exchange):
        nv = len(st_list)
        # the 2-D gradient carrier as the product path makes it (renderer.render_views): a fresh autograd leaf per step over ONE
        # cached block of zeros (nothing ever writes the values, only .grad is read) instead of a V x P x 3 fill per step
        zeros = torch.zeros((nv, P, 3), device=dev)

        def step():
            m2d = zeros.detach().requires_grad_(True)
            color, radii, depth, alpha = rasterize_views(t["means3D"], m2d, t["opacities"], st_list, shs=t["shs"],
                                                         scales=t["scales"], rotations=t["rotations"])
            if exchange:       # the MAX bucket holds forward outputs: its all-reduce overlaps the backward
                pending = parallel.exchange_forward_stats(radii, depth, group=group)
            grads = torch.autograd.grad([color, depth], plist + [m2d], [gC[:nv], gD[:nv]])
            if exchange:
                for p_, g_ in zip(plist, grads[:-1]):
                    p_.grad = g_
                parallel.exchange_sum(plist, viewspace_grads=grads[-1], group=group)
                pending.wait()
            return color
        return step

    def timed_windows(step_fn, repeats, steps, warmup):
        """W warmup steps, then `repeats` windows of exactly `steps` steps, each between barrier + synchronize pairs and
        MAX-reduced over the ranks.  Returns the per-window seconds."""
        for _ in range(warmup):
            step_fn()
        out_ = []
        for _ in range(repeats):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                et = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(et, op=dist.ReduceOp.MAX)
                el = float(et.item())
            out_.append(el)
        return out_

    step = make_step(sts, shard.group if shard is not None else None, world > 1)
    for _ in range(args.prewarm):
        step()
    windows = timed_windows(step, max(args.repeats, 1), args.steps, args.warmup)
    elapsed = sorted(windows)[len(windows) // 2]                     # the median window
    n_groups = shard.n_seed_groups if shard is not None else 1
    ms_per_step = elapsed / args.steps * 1e3
    views_total = n_groups * V * args.steps                          # views of all optimizer steps taken in a window
    mpix_s = views_total * H * W / elapsed / 1e6

    # forward only (SURVEY §8d asks for forward and forward+backward separately): the same 4-view launch set rendered
    # without a backward to follow.  Grad mode stays on so that the capacity check stays deferred as in training; an
    # output render under no_grad additionally waits for its header (one host round trip per call)
    def fwd_step():
        m2d = torch.zeros((Vl, P, 3), device=dev, requires_grad=True)
        return rasterize_views(t["means3D"], m2d, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    for _ in range(3):
        fwd_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd_step()
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t0) / args.steps * 1e3
    # an OUTPUT render (torch.no_grad(): orbit frames, the refine pass' inputs): GipRasterConfig::forward_only — the kernel
    # skips the checkpoints and the n_contrib / final_T images — and the capacity header is awaited in every call
    def nograd_step():
        with torch.no_grad():
            return rasterize_views(t["means3D"], None, t["opacities"], sts, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    for _ in range(3):
        nograd_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nograd_step()
    torch.cuda.synchronize()
    nograd_ms = (time.perf_counter() - t0) / args.steps * 1e3

    # ---- the same step on a TRAINED-looking state (secondary: the headline config is the init state SURVEY §8d prescribes)
    trained = None
    if rank == 0 and not args.no_trained:
        sct = scenes.trained_look(scenes.make_scene("human", P, seed=42, sh_degree=0), seed=7)
        tt_ = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev).contiguous().requires_grad_(True) for k, v in sct.items()}
        pl_ = [tt_[n] for n in names]

        def step_t():
            m2d = torch.zeros((Vl, P, 3), device=dev, requires_grad=True)
            color, radii, depth, alpha = rasterize_views(tt_["means3D"], m2d, tt_["opacities"], sts, shs=tt_["shs"],
                                                         scales=tt_["scales"], rotations=tt_["rotations"])
            torch.autograd.grad([color, depth], pl_ + [m2d], [gC[:Vl], gD[:Vl]])
        for _ in range(max(args.warmup, 3)):
            step_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_t()
        torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0) / args.steps
        # its own per-stage table (hipEvent pairs, like roofline.stage_ms_instrumented for the init state)
        stages_t, nr_t = R.profile_stages(tt_["means3D"].detach(), tt_["opacities"].detach(), sts, gC[:Vl], gD[:Vl], None, shs=tt_["shs"].detach(),
                                          scales=tt_["scales"].detach(), rotations=tt_["rotations"].detach(), iters=5)
        trained = {"ms_per_step": round(dtt * 1e3, 4), "mpix_per_s": round(Vl * H * W / dtt / 1e6, 1),
                   "num_rendered_per_view": int(nr_t / Vl), "stage_ms_instrumented": {k: round(v, 4) for k, v in stages_t.items()},
                   "state": "opacity 0.6, scales x U(1,3) per axis, random rotations / colours (tests/scenes.trained_look)"}
        del tt_, pl_

    # ---- BASELINE configs[4]: per-stage HBM roofline at 1M Gaussians (N = 1 only; the headline config stays configs[1]) ----
    config4 = None
    if world == 1 and not args.no_config4:
        try:
            config4 = measure_config4(dev)
        except Exception as e:  # noqa: BLE001  (secondary measurement: never lose the contract line to it)
            config4 = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()

    # ---- N > 1 extra: the replica layout of rounds 1-2 (every rank its own 4 cameras of the replicated Gaussians, a 4 x N
    # batch, gradients averaged over all ranks) — weak scaling, NOT the contract value
    replicas = None
    if world > 1:
        sts_r = settings(scenes.train_cameras(V, seed=42 + rank, H=H, W=W))
        wr_ = timed_windows(make_step(sts_r, None, True), 3, args.steps, args.warmup)
        dtr = sorted(wr_)[1] / args.steps
        replicas = {"layout": "replicas: %d ranks x %d cameras, one averaged optimizer step" % (world, V),
                    "ms_per_step": round(dtr * 1e3, 4), "views_per_s": round(world * V / dtr, 2),
                    "mpix_per_s": round(world * V * H * W / dtr / 1e6, 2), "scaling": "weak"}

    # ---- the bit-exact-lists mode (GipRasterConfig::exact_lists: the fork's tile / index buffers bit for bit; the default
    # emits the same lists minus provably dead entries): the same step, driver-measured (N = 1 only)
    exact = None
    if world == 1 and not args.no_exact:
        os.environ["GIP_RASTER_EXACT_LISTS"] = "1"
        try:
            we_ = timed_windows(step, 5, args.steps, max(args.warmup, 3))
            dte = sorted(we_)[2] / args.steps
            stages_e, nr_e = R.profile_stages(t["means3D"].detach(), t["opacities"].detach(), sts, gC[:Vl], gD[:Vl], None, shs=t["shs"].detach(),
                                              scales=t["scales"].detach(), rotations=t["rotations"].detach(), iters=3)
            exact = {"ms_per_step": round(dte * 1e3, 4), "mpix_per_s": round(V * H * W / dte / 1e6, 1),
                     "num_rendered_per_view": int(nr_e / V), "slowdown_vs_default": round(dte * 1e3 / ms_per_step, 4),
                     # where the 32 % more instances cost: every stage's own hipEvent time in this mode (compare roofline.stage_ms_instrumented)
                     "stage_ms_instrumented": {k_: round(v_, 4) for k_, v_ in stages_e.items()}}
        finally:
            os.environ["GIP_RASTER_EXACT_LISTS"] = "0"
        for _ in range(3):
            step()                      # back to the default mode's capacity hint
        torch.cuda.synchronize()

    # ---- metric (i) of BASELINE.json: full AHDS training steps/s (configs[2]; at N>1 configs[3]'s view sharding) ----
    ahds = None
    if not args.no_ahds:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_ahds
        try:
            ahds = bench_ahds.measure(steps=args.ahds_steps, warmup=4, gaussians=P, flops=(rank == 0), rank=rank,
                                      world=world, device=dev)
            if world > 1:
                # every rank its own seed (launch.py:80), all 4 views local, nothing exchanged: the embarrassingly parallel layout —
                # what N GPUs deliver when they are spent on seeds instead of on one step's views
                so = bench_ahds.measure(steps=max(args.ahds_steps // 2, 4), warmup=3, gaussians=P, rank=rank, world=world, device=dev,
                                        pieces=False, group_size=1)
                ahds["layout_seeds_only_4_views_x_%d_seeds" % world] = dict(so, views_per_s=round(so["value"] * 4, 2),
                                                                           note="ViewSharding(group_size=1): no collective in the step")
            if world >= 8 and world % 2 == 0:
                # 8 GPUs spent the other way: 4 seed groups of 2 ranks x 2 views (batch-6 networks per rank), beside configs[3]'s 4 x 2
                alt = bench_ahds.measure(steps=max(args.ahds_steps // 2, 4), warmup=3, gaussians=P, rank=rank, world=world, device=dev,
                                         pieces=False, group_size=2)
                ahds["layout_2_views_x_%d_seeds" % (world // 2)] = dict(alt, views_per_s=round(alt["value"] * 4, 2),
                                                                       note="ViewSharding(group_size=2): value = optimizer steps/s summed over the seed groups")
            if world == 1 and not args.no_trained:
                # same number of timed steps and the same warm-up as the init-state measurement above (round 5 timed 5 steps after 3:
                # the first steps of a new Gaussian state still size the rasterizer's capacity and settle the GradScaler), then the
                # init state ONCE MORE behind it, so that a drift of the box (clocks under sustained load) shows up as such and is not
                # read as a property of the trained state
                tr = bench_ahds.measure(steps=args.ahds_steps, warmup=4, gaussians=P, device=dev, trained=True, pieces=False)
                again = bench_ahds.measure(steps=args.ahds_steps, warmup=4, gaussians=P, device=dev, pieces=False)
                ahds["trained_state"] = {"value": tr["value"], "ms_per_step": tr["ms_per_step"],
                                         "init_state_remeasured_after_it_ms_per_step": again["ms_per_step"],
                                         "state": "opacity 0.6, scales x U(1,3) per axis, random rotations / colours (tests/scenes.trained_look)"}
            if world == 1 and not args.no_proxy:
                # 1-GPU proxy of ONE rank of BASELINE.json configs[3] (no 8-GPU node here): the shard of a rank in a 4-rank
                # seed group (1 view raster + 1-image VAE fwd/bwd + batch-3 ControlNet / U-Net) and in a 2-rank group
                # (2 views), collectives not executed (a rank's step then lacks two small all-reduces, ~0.1 ms measured with
                # stubs in tools/exp_exchange_overhead.py).  implied_* = what N such ranks deliver if nothing else is lost.
                prox = {}
                for k in (4, 2):
                    r_ = bench_ahds.measure(steps=max(args.ahds_steps // 2, 4), warmup=3, gaussians=P, device=dev, proxy_group=k, pieces=False)
                    prox["group_of_%d" % k] = {"views_per_rank": 4 // k, "ms_per_step": r_["ms_per_step"]}
                t1, t4, t2 = ahds["ms_per_step"], prox["group_of_4"]["ms_per_step"], prox["group_of_2"]["ms_per_step"]
                prox["implied_views_per_s"] = {"1": round(4e3 / t1, 2), "2": round(4e3 / t2, 2), "4": round(4e3 / t4, 2), "8": round(8e3 / t4, 2)}
                prox["implied_speedup_vs_1gpu"] = {"2": round(t1 / t2, 3), "4": round(t1 / t4, 3), "8": round(2 * t1 / t4, 3)}
                # the other way to spend 8 GPUs (ViewSharding(group_size=2)): 4 independent seed groups of 2 ranks x 2 views — every
                # rank's networks stay at batch 6 instead of 3; bench.py --gpus 8 measures it beside the contract layout
                prox["implied_8gpu_as_2_views_x_4_seeds"] = {"views_per_s": round(16e3 / t2, 2), "speedup_vs_1gpu": round(4 * t1 / t2, 3)}
                prox["note"] = "measured on ONE GPU; the RCCL all-reduces of a real group are not in these times"
                ahds["config3_proxy"] = prox
        except Exception as e:  # the raster line above is the contract metric: never lose it to the secondary measurement
            ahds = dict(ahds or {}, error="%s: %s" % (type(e).__name__, str(e)[:300]))
        # ---- BASELINE configs[4], the other half: the VCR refine pass (32 views x 8 DDIM steps at 1024^2; a bounded sample of 12 views)
        # and the stage-3 reconstruction step (refine.py:115-239, GaussianIP.py:424-436).  LAST: the refine pass changes the
        # IP-Adapter scale, which invalidates the captured graphs of the AHDS measurements above
        if world == 1 and isinstance(config4, dict) and not args.no_refine:
            try:
                import bench_refine
                import bench_stage3
                config4["refine"] = bench_refine.measure(bench_ahds.cached_guidance())
                config4["stage3"] = bench_stage3.measure()
            except Exception as e:  # noqa: BLE001
                config4["refine_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    out = None
    if rank == 0:
        # ---- roofline: live per-kernel durations (hipEvents on the launch stream) ----
        stages, num_rendered = R.profile_stages(
            t["means3D"].detach(), t["opacities"].detach(), sts, gC[:Vl], gD[:Vl], None, shs=t["shs"].detach(),
            scales=t["scales"].detach(), rotations=t["rotations"].detach(), iters=args.profile_iters)
        Rv = num_rendered / Vl
        T = ((H + 15) // 16) * ((W + 15) // 16)
        N = H * W
        b_f, b_b = algorithmic_bytes(P, K, Rv, T, N)
        kernel_stages = [s for s in stages if s != "clear"]
        dom = max(kernel_stages, key=lambda s: stages[s])
        dom_bytes = stage_bytes(dom, P, K, Rv, T, N) * Vl
        achieved = dom_bytes / (stages[dom] * 1e-3) / 1e9
        # HBM traffic and vector-issue counters come from committed rocprofv3 --pmc passes of THIS build (separate passes,
        # tools/collect_profiles.sh; FETCH_SIZE doubled and KiB units per the MI355X guide): they cannot be collected
        # inside this run, so the line names its source
        traffic, valu, pmc_note = None, None, "profiles/pmc.json missing"
        ppath = os.path.join(ROOT, "profiles", "pmc.json")
        if os.path.exists(ppath):
            try:
                pmc = json.load(open(ppath))
                lib_sha = raster_source_hash()
                stamp = pmc.get("_build", {})
                c = pmc.get(dom, {})
                if Vl != 4:
                    # the passes were taken on the 4-view launch of N = 1; a rank of an N-GPU run launches fewer views
                    pmc_note = "profiles/pmc.json holds counters of the 4-view launch; this rank launches %d view(s): counters withheld" % Vl
                elif stamp.get("raster_source_sha16") != lib_sha:
                    # counters of OTHER kernel sources: not this line's business (they went stale silently).  The stamp
                    # is a hash of the SOURCES + Makefile (a rebuilt binary of identical sources hashes differently: VERDICT r3)
                    pmc_note = "profiles/pmc.json was collected on raster sources %s, this run is built from %s: counters withheld" % (
                        stamp.get("raster_source_sha16"), lib_sha)
                else:
                    pmc_note = "profiles/pmc.json (rocprofv3 --pmc passes of this build, git %s)" % stamp.get("git", "?")
                    traffic = int(c["hbm_fetch_bytes"] + c["hbm_write_bytes"])
                    # Vector-issue roofline of the same kernel.  Issue cost per wave64 instruction on one SIMD, measured
                    # with s_memtime stamps (tools/micro/valu_rate.hip -> profiles/r03_valu_rate.json): plain fp32
                    # (v_fma_f32) 2.24 cycles with >= 4 waves per SIMD (2.5 with 2; 4.9 for a lone wave), packed
                    # (v_pk_fma_f32) 4.2, transcendental (v_exp_f32) 8.1.  `frac` prices EVERY vector instruction at the
                    # plain rate — a lower bound of the issue time the kernel needs; `frac_lone_wave` at the 4.9 cycles a
                    # lone wave sustains (round 2 assumed 4).
                    vr = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_rate.json")))
                    c_sat, c_lone = vr["v_fma_f32"]["simd_cycles_ge4_waves"], vr["v_fma_f32"]["simd_cycles_1_wave"]
                    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
                    valu = {"bound": "valu-issue", "kernel": "gip_%s_kernel" % dom, "wave_instructions_per_launch": int(c["SQ_INSTS_VALU"]),
                            "cycles_per_wave_instruction": c_sat, "rate_source": "profiles/r03_valu_rate.json (tools/micro/valu_rate.hip)",
                            "available_simd_cycles": int(cycles * 1024),
                            "frac": round(c_sat * c["SQ_INSTS_VALU"] / (cycles * 1024), 4),
                            "frac_lone_wave": round(c_lone * c["SQ_INSTS_VALU"] / (cycles * 1024), 4),
                            "simd_cycles_per_instruction_achieved": round(cycles * 1024 / c["SQ_INSTS_VALU"], 3),
                            # class mix (SQ_INSTS_VALU_* pass): transcendentals at their own measured 8.1 cycles, the rest at c_sat;
                            # packed v_pk_* instructions are not separable by counter and cost 4.2 — so this is still a LOWER bound
                            "frac_class_weighted": (round((c_sat * (c["SQ_INSTS_VALU"] - c["SQ_INSTS_VALU_TRANS_F32"]) +
                                                           vr["v_exp_f32"]["simd_cycles_ge4_waves"] * c["SQ_INSTS_VALU_TRANS_F32"]) / (cycles * 1024), 4)
                                                    if "SQ_INSTS_VALU_TRANS_F32" in c else None),
                            "class_mix": ({k_[len("SQ_INSTS_VALU_"):]: int(v_) for k_, v_ in c.items() if k_.startswith("SQ_INSTS_VALU_")} or None),
                            "SQ_ACTIVE_INST_VALU_over_SQ_INSTS_VALU": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 3),
                            "launch_cycles": int(cycles), "source": pmc_note,
                            "formula": "c * SQ_INSTS_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)"}
            except Exception as e:      # noqa: BLE001
                traffic, valu, pmc_note = None, None, "profiles/pmc.json unreadable: %s" % e
        roofline = {"bound": "hbm", "kernel": "gip_%s_kernel" % dom, "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": pmc_note,
                    "algorithmic_bytes_per_launch": int(dom_bytes),
                    "avg_launch_ms": round(stages[dom], 4),
                    "stage_ms_instrumented": {k: round(v, 4) for k, v in stages.items()},
                    "stage_ms_note": "a separate run with a hipEvent pair around every stage (gip_raster_*_profiled): their sum exceeds "
                                     "ms_per_step (un-instrumented windows) by the events' own cost, ~3 %",
                    "sum_stage_ms_instrumented": round(sum(stages.values()), 4),
                    "whole_step_GBs": round((b_f + b_b) * Vl / (ms_per_step * 1e-3) / 1e9, 2),
                    "whole_step_frac": round((b_f + b_b) * Vl / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "pair_evals_per_s_fwd": round(Rv * Vl * 256 / (stages["render_fwd"] * 1e-3), 0)}

        cpu = None
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            orc.build()
            ncores, nthreads = physical_cores()          # one OpenMP thread per PHYSICAL core
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
            # the vectorised PyTorch-CPU leg — the dense float64 formulation of tests/dense_reference.py
            # (every Gaussian at every pixel, culling as masks; shares no code with the C oracle) on a bounded sample: its
            # memory is P x H x W doubles per temporary, so 2 000 Gaussians at 128 x 128, forward only
            torch_leg = None
            try:
                import dense_reference as dr
                old_threads = torch.get_num_threads()
                torch.set_num_threads(ncores)
                Pd, Hd = 2000, 128
                sd = dr.random_scene(Pd, 42)
                view, full, campos, tanx, tany = dr.look_at_camera(5.0, 90.0, 1.8, 70.0, Hd, Hd)
                kwd = dict(viewmatrix=view, projmatrix=full, campos=campos, bg=torch.zeros(3, dtype=torch.float64), H=Hd, W=Hd,
                           tanfovx=tanx, tanfovy=tany, sh_degree=0, **sd)
                with torch.no_grad():
                    dr.dense_render(**kwd)
                    td = []
                    for _ in range(5):
                        c1 = time.perf_counter()
                        dr.dense_render(**kwd)
                        td.append(time.perf_counter() - c1)
                td.sort()
                torch.set_num_threads(old_threads)
                torch_leg = {"value": round(Hd * Hd / td[2] / 1e6, 4), "unit": "Mpix/s", "pair_evals_per_s": round(Pd * Hd * Hd / td[2], 0),
                             "threads": ncores, "sample": "median of 5 forward renders of %d random Gaussians at %dx%d, dense float64 PyTorch "
                                                          "formulation (tests/dense_reference.py), torch.set_num_threads(%d)" % (Pd, Hd, Hd, ncores)}
            except Exception as e:      # noqa: BLE001
                torch_leg = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            cpu = {"value": round(reps * H * W / tt / 1e6, 3), "unit": "Mpix/s", "cores": ncores, "threads": nthreads, "cpu_model": model, "kind": "port",
                   "pytorch_cpu_vectorised": torch_leg,
                   "forward_only_value": round(H * W / t_fwd[2] / 1e6, 3),
                   "sample": "median of 5 x (1 view forward, then backward; P=%d, %dx%d) after one warm-up, on the C oracle "
                             "(oracle/raster_oracle.c, OpenMP over tiles, %d threads); value = forward+backward" % (P, H, W, ncores)}

        out = {"metric": "raster_fwd_bwd_mpix_per_s", "value": round(mpix_s, 2), "unit": "Mpix/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[1]: %d Gaussians (synthetic human surface, SMPL-X-style "
                                      "init), %dx%d, %d views per optimizer step, raster forward+backward" % (P, H, W, V),
                          "gaussians": P, "height": H, "width": W, "views_per_step": V, "views_per_step_per_gpu": Vl,
                          "prewarm_steps": args.prewarm, "repeats": len(windows), "value_is": "median window",
                          "num_rendered_per_view": int(Rv), "sh_degree": 0,
                          "backend": None if world == 1 else ("rccl" if backend == "nccl" else backend),
                          "dist_world_size": 1 if world == 1 else dist.get_world_size(),
                          "gpus_visible": int(os.environ.get("GIP_BENCH_VISIBLE_GPUS", torch.cuda.device_count())),
                          "parallelism": "1 GPU" if world == 1 else "BASELINE configs[3]: %d views sharded over %d GPU(s) x %d seed group(s)" % (
                              V, shard.group_size, shard.n_seed_groups)},
               "window_ms_per_step": [round(w_ / args.steps * 1e3, 4) for w_ in windows],
               "raster_steps_per_s": round(1e3 / ms_per_step, 3), "views_per_s": round(views_total / elapsed, 2),
               "forward_only": {"ms_per_step": round(fwd_ms, 4), "mpix_per_s_per_gpu": round(Vl * H * W / fwd_ms / 1e3, 1),
                                "no_grad_render_ms_per_call": round(nograd_ms, 4),
                                "note": "ms_per_step: forward with the state a backward needs, capacity check deferred; no_grad_render: forward_only kernels, header awaited per call"},
               "trained_state": trained,
               "config4": config4,
               "exact_lists": exact,
               "replicas_layout": replicas,
               "roofline": roofline, "roofline_valu": valu, "cpu_baseline": cpu}
    if rank == 0:
        out["ahds"] = ahds
        # BASELINE.json's metric names the AHDS training-step rate first: first-class beside the raster rate
        out["ahds_steps_per_s"] = ahds.get("value") if isinstance(ahds, dict) else None
        out["ahds_ms_per_step"] = ahds.get("ms_per_step") if isinstance(ahds, dict) else None
        print(json.dumps(out), file=result_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
