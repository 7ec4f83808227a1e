import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, networks, refine as rf
from gaussianip_amd.guidance.ahds import AHDSSchedule
gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
dec = networks.init_for_benchmark(networks.VAEDecoder(), 5).to("cuda", torch.float16).eval().requires_grad_(False).to(memory_format=torch.channels_last)
vcr = rf.ViewConsistentRefiner(gd, dec)
g = torch.Generator(device="cuda").manual_seed(0)
rgb = torch.rand(32, 1024, 1024, 3, device="cuda", generator=g); ctrl = torch.rand(32, 1024, 1024, 3, device="cuda", generator=g)
cond = torch.randn(1, 77, 768, device="cuda", generator=g) * 0.1; uncond = torch.randn(1, 77, 768, device="cuda", generator=g) * 0.1
fn = lambda n: (cond, uncond)
vcr.refine_rgb(rgb, ctrl, fn, views=["front"])
vcr.refine_rgb(rgb, ctrl, fn, views=["front", "k0", "v3"])
torch.cuda.synchronize()
