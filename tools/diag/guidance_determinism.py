"""Run-to-run reproducibility of the guidance pieces at the shapes of a 2-view shard (denoise batch 6, VAE batch 2) and of the
4-view step (12 / 4): same inputs, same generator seeds, repeated calls -> bitwise equal?  (The sharded configs[3] test found
its losses moving between runs.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance  # noqa: E402
from gaussianip_amd.guidance.prompts import PromptProcessor  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev), torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
gd = StableDiffusionGuidance(GuidanceConfig(), image_embeds_provider=lambda _: tokens)
pp = PromptProcessor("a person wearing a coat", lambda texts: torch.randn(len(texts), 77, 768, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).half() * 0.1,
                     negative_prompt="blurry")
gd.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)
pu = pp()


def inputs(B, seed):
    gg = torch.Generator(device=dev).manual_seed(seed)
    lat = torch.randn(B, 4, 64, 64, device=dev, generator=gg)
    t = torch.randint(20, 800, (B,), device=dev, generator=gg)
    ctrl = torch.rand(B, 3, 512, 512, device=dev, generator=gg)
    emb = (torch.randn(3 * B, 81, 768, device=dev, generator=gg) * 0.1).half()
    return torch.cat([lat] * 3), ctrl, torch.cat([t] * 3), emb


for B in (2, 4, 1):
    x, ctrl, t, emb = inputs(B, 0)
    x2, ctrl2, t2, emb2 = inputs(B, 1)
    outs = []
    with torch.no_grad():
        for i in range(6):
            outs.append(gd.forward_unet(x, ctrl, t, emb, True, replicas=3).clone())
            gd.forward_unet(x2, ctrl2, t2, emb2, True, replicas=3)
    print("forward_unet B=%d x3: calls equal to call 0: %s   max |diff| %s" % (
        B, [bool(torch.equal(o, outs[0])) for o in outs], ["%.2e" % float((o - outs[0]).abs().max()) for o in outs]), flush=True)
    os.environ["GIP_GUIDANCE_STREAMS"] = "2"
    img = torch.rand(B, 3, 512, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    G = torch.randn(B, 4, 64, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) * 0.03
    zs, gs = [], []
    for i in range(6):
        xi = img.clone().requires_grad_(True)
        z = gd.encode_images(xi, torch.Generator(device=dev).manual_seed(77))
        (z * G).sum().mul(1024.0).backward()
        zs.append(z.detach().clone())
        gs.append(xi.grad.clone())
    print("encode_images B=%d: latents equal %s grads equal %s   max |dgrad| %s" % (
        B, [bool(torch.equal(o, zs[0])) for o in zs], [bool(torch.equal(o, gs[0])) for o in gs], ["%.2e" % float((o - gs[0]).abs().max()) for o in gs]), flush=True)
    # the whole plugin call
    rgb = torch.rand(B, 1024, 1024, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    pose = torch.rand(B, 512, 512, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    el = torch.linspace(-20, 20, B)
    az = torch.linspace(-150, 120, B)
    losses, grads = [], []
    for i in range(5):
        r = rgb.clone().requires_grad_(True)
        gens = [torch.Generator(device=dev).manual_seed(1000 + v) for v in range(B)]
        o = gd(750, r, pose, pu, True, torch.ones(B, device=dev), elevation=el, azimuth=az, center=torch.zeros(B), camera_distances=torch.full((B,), 1.5), generator=gens)
        (o["loss_sds"] * 1024.0).backward()
        losses.append(float(o["loss_sds"]))
        grads.append(r.grad.clone())
    print("guidance() B=%d: loss_sds %s   grads equal to call 0 %s rel L2 %s" % (
        B, ["%.6f" % v for v in losses], [bool(torch.equal(o, grads[0])) for o in grads],
        ["%.2e" % float((o - grads[0]).double().norm() / grads[0].double().norm()) for o in grads]), flush=True)
