#!/usr/bin/env python3
"""VCR refine pass at full size (configs[4]: 1024^2 renders -> 128^2 latents, 8 DDIM steps, CFG batch 2): seconds per
view for each attention branch, the projected time for the 32-view pass, and — since round 6 — the FLOPs of a canonical view
(VAE encode + 8 x (ControlNet + U-Net at batch 2) + VAE decode, counted on the plain-PyTorch path) against the fp16 dense peak.
Random-initialised networks.  `measure()` is also the `config4.refine` object of bench.py (a bounded sample: 12 of the 32 views)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(gd=None, count_flops=True):
    import torch
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, fused, networks, refine as rf
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    if gd is None:
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
    vcr.refine_rgb(rgb, ctrl, fn, views=["front"])           # warm-up (allocator, graph-less eager path)
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
    out = {"workload": "VCR refine (refine.py:115-239, pipeline_ipa_controlnet.py:1447-1877), 1024^2 renders -> 128^2 latents, 8 DDIM steps, CFG 7.5, fp16",
           "sample": "12 of the 32 views timed (4 canonical, 4 key views with mutual self-attention, 4 blended views); the pass is projected from them",
           "s_per_view": {"canonical": round(per_canon, 3), "key_mutual": round(per_key, 3), "blend": round(per_blend, 3)},
           "projected_32_views_s": round(4 * per_canon + 4 * per_key + 24 * per_blend, 2), "raw_s": res}
    if count_flops:
        # FLOPs of ONE canonical view on the plain-PyTorch path (the counter cannot see HIP launches): the key / blended views add
        # attention over 2 N / 3 x N keys on top, so pricing every view at this count UNDER-states their rate
        from torch.utils.flop_counter import FlopCounterMode
        try:
            with fused.disabled(), FlopCounterMode(display=False) as fc:
                vcr.refine_rgb(rgb, ctrl, fn, views=["front"])
            fl = fc.get_total_flops()
            out["flops_per_canonical_view"] = int(fl)
            out["canonical_tflops_per_s"] = round(fl / per_canon / 1e12, 1)
            out["canonical_mfma_frac"] = round(fl / per_canon / 2.5e15, 4)
            out["pass_lower_bound_mfma_frac"] = round(32 * fl / out["projected_32_views_s"] / 2.5e15, 4)
        except Exception as e:      # noqa: BLE001
            out["flops_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
    del vcr, dec, rgb, ctrl
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(measure()), flush=True)
