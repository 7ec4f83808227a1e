import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    vcr.refine_rgb(rgb, ctrl, fn, views=["front"])
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if ("conv" in e.key.lower() and "conv3x3_kernel" not in e.key)]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print("%9.3f ms  %4d  %-60s %s" % (e.device_time_total / 1e3, e.count, e.key[:60], str(e.input_shapes)[:120]))
