"""Is the denoise bit-reproducible run to run?  (one stream vs two streams, small and full-size latents)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, ipa_guidance
from gaussianip_amd.guidance.ahds import AHDSSchedule
gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
g = torch.Generator(device="cuda").manual_seed(8)
for size, B in ((32, 2), (64, 4)):
    lat = torch.randn(B, 4, size, size, device="cuda", generator=g)
    ctrl = torch.rand(B, 3, size * 8, size * 8, device="cuda", generator=g)
    emb = (torch.randn(3 * B, 81, 768, device="cuda", generator=g) * 0.1).half()
    tt = torch.randint(20, 900, (B,), device="cuda", generator=g)
    x, t3 = torch.cat([lat] * 3), torch.cat([tt] * 3)
    for two in (False, True):
        ipa_guidance._TWO_STREAMS = two
        with torch.no_grad():
            outs = [gd.forward_unet(x, ctrl, t3, emb, True, replicas=3).clone() for _ in range(4)]
        torch.cuda.synchronize()
        print("latents %d^2 batch %d, two streams %s: max diff between runs %s" % (size, 3 * B, two, [float((o - outs[0]).abs().max()) for o in outs[1:]]), flush=True)
    ref1 = outs
