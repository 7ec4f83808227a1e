"""Child process of tests/test_gpu_sharded_step.py::test_config3_real_guidance_*: rank `r` of a 2-process gloo group, both
ranks on cuda:0.  BASELINE.json configs[3] at its real size and with the REAL guidance: 100 000 Gaussians, 1024 x 1024,
batch 4, StableDiffusionGuidance (VAE encode -> AHDS timestep -> ANPG over ControlNet + U-Net -> SDS loss), driven through
StageOneStep.training_step / optimizer_step with a GradScaler.  One optimizer step is run twice on identical models — once
unsharded (all 4 views on this rank, no exchange: the single-process step of configs[2]) and once view-sharded
(parallel.ViewSharding: this rank renders views r and r + 2, guidance at batch 6) — and the reduced gradients, statistics
and losses are compared.  Every view draws its VAE noise, its timestep and its SDS noise from a generator seeded by
(step, view id), so what a view sees does not depend on which other views share its batch."""
import json
import os
import sys
from argparse import ArgumentParser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    P = int(sys.argv[5]) if len(sys.argv) > 5 else 100000
    H = W = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
    strong = len(sys.argv) > 7 and sys.argv[7] == "strong"       # tests/conditioning.py: a conditioning that matters (meaningful fp16 floor)
    import numpy as np
    import torch
    import torch.distributed as dist
    import scenes
    from gaussianip_amd import parallel
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    from gaussianip_amd.guidance.prompts import PromptProcessor
    from gaussianip_amd.poser import Skeleton
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.system import StageOneStep
    from gaussianip_amd.utils import BasicPointCloud
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B = 4
    pts = scenes.human_points(P, np.random.default_rng(42)).astype(np.float32)
    g = torch.Generator(device=dev).manual_seed(1)
    tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    guidance = StableDiffusionGuidance(GuidanceConfig(), image_embeds_provider=lambda gd: tokens)        # same seeds on every rank

    def encode(texts):
        gg = torch.Generator(device=dev).manual_seed(7)
        return torch.randn(len(texts), 77, 768, device=dev, generator=gg).half() * 0.1
    pp = PromptProcessor("a person wearing a coat", encode, negative_prompt="blurry")
    guidance.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)
    prompt_utils = pp()
    if strong:
        from conditioning import strengthen_conditioning
        strengthen_conditioning(guidance)

    def guided(step, rgb, control, pu, use_pose, all_vis_all, view_id=None, **batch):
        gens = [torch.Generator(device=dev).manual_seed(100000 + 1000 * int(step) + int(v)) for v in view_id.tolist()]
        return guidance(step, rgb, control, pu, use_pose, all_vis_all, generator=gens, **batch)

    def guided_split(step, rgb, control, pu, use_pose, all_vis_all, view_id=None, **batch):
        """The single-process step with the guidance called once per SHARD of views ([0, 2] then [1, 3]): the networks then run
        the shapes the sharded ranks run (VAE batch 2, denoise batch 6), so this run and the sharded run differ only by the
        exchange — they must agree to float32 summation order.  (The one-call 4-view run uses other kernels per layer — tile
        counts, split-K factors, Winograd choices depend on the batch — i.e. other fp16 roundings, which ANPG's difference of
        near-equal noise predictions, times 7.5, amplifies to per-cent level in the gradients.)"""
        n = rgb.shape[0]
        total = 0.0
        for ids in ([0, 2], [1, 3]):
            idx = torch.as_tensor(ids, device=rgb.device)
            sub = {k: (v[torch.as_tensor(ids, device=v.device)] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n else v) for k, v in batch.items()}
            o = guided(step, rgb[idx], control[idx], pu, use_pose, all_vis_all[idx.to(all_vis_all.device)], view_id=view_id[torch.as_tensor(ids)], **sub)
            total = total + o["loss_sds"] * (len(ids) / float(n))
        return {"loss_sds": total, "grad_norm": total.detach()}

    def run(sharding, step=750, split=False):
        gm = GaussianModel(0)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
        gm.training_setup(OptimizationParams(ArgumentParser()), fused=True)
        skel = Skeleton(dev)
        skel.scale(-10)
        stage = StageOneStep(gm, PipelineParams(ArgumentParser()), torch.zeros(3, device=dev), skeleton=skel)
        stage.sharding = sharding
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=1 << 30)
        batch = scenes.train_batch(np.random.default_rng(7), B, H, W, device=None)
        batch["view_id"] = torch.arange(B)
        loss, out, gout = stage.training_step(step, batch, guided_split if split else guided, prompt_utils, True)
        stage.optimizer_step(loss, step, scaler=scaler, exchange=None if sharding is None else sharding.exchange)
        torch.cuda.synchronize()
        rec = {"grads": [g_["params"][0].grad.detach().clone() for g_ in gm.optimizer.param_groups],
               "radii": stage.radii.clone(), "accum": gm.xyz_gradient_accum.clone(), "denom": gm.denom.clone(),
               "loss": loss.detach().double().clone(), "loss_sds": gout["loss_sds"].detach().double().clone(),
               "scale": scaler.get_scale(),
               "state": [g_["params"][0].detach().clone() for g_ in gm.optimizer.param_groups],
               "calls": dict(guidance_batch=int(out["comp_rgb"].shape[0]))}
        return rec

    full = run(None)                    # configs[2]: one guidance call over the 4 views (denoise batch 12)
    ref = run(None, split=True)         # the same step with the guidance called per shard of views (denoise batch 6, twice)
    sh = run(parallel.ViewSharding(B))
    res = {"rank": rank, "P": P, "size": H, "views_local": sh["calls"]["guidance_batch"], "views_ref": ref["calls"]["guidance_batch"],
           "scale_ref": ref["scale"], "scale_sharded": sh["scale"], "scale_full": full["scale"]}
    res["radii_equal"] = bool(torch.equal(ref["radii"], sh["radii"]) and torch.equal(full["radii"], sh["radii"]))
    res["denom_equal"] = bool(torch.equal(ref["denom"], sh["denom"]))

    def rel(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return (float((a - b).norm() / b.norm().clamp_min(1e-300)), float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300)))
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]

    def table(r_):
        return {n: dict(zip(("rel_l2", "cosine"), rel(b, a))) | {"ref_norm": float(a.double().norm()), "finite": bool(torch.isfinite(b).all())}
                for n, a, b in zip(names, r_["grads"], sh["grads"]) if a.numel()}
    res["grad"] = table(ref)                     # sharded against the same-shapes single-process step: the exchange
    res["grad_vs_one_call"] = table(full)        # sharded against the one-call 4-view step: + the networks' batch-dependent fp16 roundings
    res["accum"] = dict(zip(("rel_l2", "cosine"), rel(sh["accum"], ref["accum"])))
    res["accum_vs_one_call"] = dict(zip(("rel_l2", "cosine"), rel(sh["accum"], full["accum"])))
    res["loss_one_call"] = float(full["loss"])
    # the sharded loss is this rank's SHARE: the group's sum is the batch loss
    tot = sh["loss"].clone().cpu()
    dist.all_reduce(tot)
    res["loss_ref"], res["loss_sharded_sum"] = float(ref["loss"]), float(tot)
    chk = torch.stack([t.double().sum() for t in sh["state"] if t.numel()]).cpu()
    both = [None] * world
    dist.all_gather_object(both, chk)
    res["ranks_agree_after_adam"] = bool(all(torch.equal(both[0], b) for b in both))
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
