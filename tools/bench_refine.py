#!/usr/bin/env python3
"""VCR refine pass at full size (configs[4]: 1024^2 renders -> 128^2 latents, 8 DDIM steps, CFG batch 2): seconds per
view for each attention branch and the projected time for the 32-view pass.  Random-initialised networks."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, networks, refine as rf
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
    dec = networks.init_for_benchmark(networks.VAEDecoder(), 5).to("cuda", torch.float16).eval().requires_grad_(False)
    dec = dec.to(memory_format=torch.channels_last)
    vcr = rf.ViewConsistentRefiner(gd, dec)
    g = torch.Generator(device="cuda").manual_seed(0)
    H = W = 1024
    rgb = torch.rand(32, H, W, 3, device="cuda", generator=g)
    ctrl = torch.rand(32, H, W, 3, device="cuda", generator=g)
    cond = torch.randn(1, 77, 768, device="cuda", generator=g) * 0.1
    uncond = torch.randn(1, 77, 768, device="cuda", generator=g) * 0.1
    fn = lambda n: (cond, uncond)  # noqa: E731
    vcr.refine_rgb(rgb, ctrl, fn, views=["front"])           # warm-up (MIOpen find, allocator)
    res = {}
    for label, views in (("canonical", ["front", "back", "left", "right"]), ("key_mutual", ["front", "back", "left", "right", "k0", "k1", "k2", "k3"]),
                         ("blend", ["front", "back", "left", "right", "k0", "k1", "k2", "k3", "v0", "v1", "v2", "v3"])):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _ = vcr.refine_rgb(rgb, ctrl, fn, views=views)
        torch.cuda.synchronize()
        res[label] = round(time.perf_counter() - t0, 3)
    per_canon = res["canonical"] / 4
    per_key = (res["key_mutual"] - res["canonical"]) / 4
    per_blend = (res["blend"] - res["key_mutual"]) / 4
    print(json.dumps({"workload": "VCR refine, 1024^2, 8 DDIM steps, CFG 7.5, fp16", "s_per_view": {"canonical": round(per_canon, 3), "key_mutual": round(per_key, 3), "blend": round(per_blend, 3)},
                      "projected_32_views_s": round(4 * per_canon + 4 * per_key + 24 * per_blend, 2), "raw": res}), flush=True)


if __name__ == "__main__":
    main()
