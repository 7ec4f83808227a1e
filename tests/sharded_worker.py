"""Child process of tests/test_gpu_sharded_step.py: rank `r` of a 2-process gloo group, both ranks on cuda:0.
Runs two optimizer steps (the second one densifies) of the stage-1 step twice on identical models — once unsharded
(all 4 views on this rank, no exchange) and once view-sharded (parallel.ViewSharding: this rank renders views r, r + 2) —
and compares the exchanged statistics, the reduced gradients and the post-densify state."""
import json
import os
import sys
from argparse import ArgumentParser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import numpy as np
    import torch
    import torch.distributed as dist
    import scenes
    from gaussianip_amd import parallel
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.system import StageOneConfig, StageOneStep
    from gaussianip_amd.utils import BasicPointCloud
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    P, H, W, B = 20000, 256, 256, 4
    pts = scenes.human_points(P, np.random.default_rng(42)).astype(np.float32)
    K = torch.randn(B, H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)     # per-view loss weights, same on every rank

    def guidance(step, rgb, control, prompt_utils, use_pose, all_vis_all, view_id=None, **batch):
        """A deterministic differentiable stand-in with the plugin's signature: a per-view functional of the render,
        averaged over the views it is given (like loss_sds / batch_size)."""
        k = K[view_id.to(dev)]
        per_view = (rgb * k).sum(dim=(1, 2, 3)) + 0.05 * (rgb ** 2).sum(dim=(1, 2, 3))
        return {"loss_sds": per_view.sum() / rgb.shape[0], "grad_norm": per_view.detach().norm()}

    def make():
        gm = GaussianModel(0)
        gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():               # anisotropic, rotated splats: every parameter group gets a meaningful gradient
            gm._scaling += (torch.randn(P, 3, generator=g) * 0.4).to(dev)
            gm._rotation += (torch.randn(P, 4, generator=g) * 0.5).to(dev)
        gm.training_setup(OptimizationParams(ArgumentParser()))
        cfg = StageOneConfig(densify_prune_start_step=0, densify_prune_interval=2, max_grad=2e-5)
        return gm, StageOneStep(gm, PipelineParams(ArgumentParser()), torch.zeros(3, device=dev), cfg)

    def run(sharding):
        gm, stage = make()
        stage.sharding = sharding
        rng = np.random.default_rng(7)
        rec = {}
        for step in (1, 2):
            batch = scenes.train_batch(rng, B, H, W, device=dev)
            batch.pop("mvp_mtx")
            batch["view_id"] = torch.arange(B)
            out = stage.forward(batch)          # (training_step needs pose maps; the raster + loss + exchange path is what is compared)
            ids = list(range(B)) if sharding is None else list(sharding.views)
            g_out = guidance(step, out["comp_rgb"], None, None, True, None, view_id=batch["view_id"][ids])
            loss = stage.loss(out, g_out) * (1.0 if sharding is None else sharding.share)
            torch.manual_seed(1000 + step)      # the split samples of densify_and_prune come from the global generator
            action = stage.optimizer_step(loss, step, exchange=None if sharding is None else sharding.exchange)
            if step == 1:
                rec["grads"] = [g_["params"][0].grad.detach().clone() for g_ in gm.optimizer.param_groups]
                rec["radii"] = stage.radii.clone()
                rec["accum"] = gm.xyz_gradient_accum.clone()
                rec["opacity_scale"] = float((out["depth"] / out["opacity"].clamp_min(1e-30)).max())
            else:
                assert action == "densify_and_prune", action
        rec["state"] = [g_["params"][0].detach().clone() for g_ in gm.optimizer.param_groups]
        rec["lrs"] = [float(g_["lr"]) for g_ in gm.optimizer.param_groups]
        rec["count"] = int(gm.get_xyz.shape[0])
        return rec

    ref = run(None)
    sh = run(parallel.ViewSharding(B))
    torch.cuda.synchronize()
    res = {"rank": rank, "count_ref": ref["count"], "count_sharded": sh["count"]}
    res["radii_equal"] = bool(torch.equal(ref["radii"], sh["radii"]))                       # ints: bitwise
    res["depth_max_rel"] = abs(ref["opacity_scale"] - sh["opacity_scale"]) / ref["opacity_scale"]
    res["grad_rel_each"] = [float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) if a.numel() else 0.0 for a, b in zip(ref["grads"], sh["grads"])]
    res["grad_ratio"] = [float(b.abs().sum() / a.abs().sum().clamp_min(1e-30)) if a.numel() else 0.0 for a, b in zip(ref["grads"], sh["grads"])]
    res["grad_rel"] = max(res["grad_rel_each"])
    res["accum_rel"] = float((ref["accum"] - sh["accum"]).abs().max() / ref["accum"].abs().max())
    if ref["count"] == sh["count"]:
        # Adam's first steps move a parameter by ~lr * sign(gradient): an element whose gradient is zero up to rounding may
        # step the other way (2 lr per step apart); everything else must agree to rounding
        diffs = [(a - b).abs() for a, b in zip(ref["state"], sh["state"]) if a.numel()]
        lrs = [lr for lr, a in zip(ref["lrs"], ref["state"]) if a.numel()]
        res["state_mismatch_frac"] = max(float((d > 1e-6).float().mean()) for d in diffs)
        res["state_max_over_lr"] = max(float(d.max()) / lr for d, lr in zip(diffs, lrs))
    # both ranks must hold the same sharded state
    chk = torch.stack([t.double().sum() for t in sh["state"] if t.numel()]).cpu()
    both = [None] * world
    dist.all_gather_object(both, chk)
    res["ranks_agree"] = bool(all(torch.equal(both[0], b) for b in both))
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
