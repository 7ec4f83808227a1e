import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
from gaussianip_amd.guidance.ahds import AHDSSchedule
dev = torch.device("cuda")
g = StableDiffusionGuidance(GuidanceConfig(channels_last=(os.environ.get("CL","0")=="1")), schedule=AHDSSchedule(list(range(2400))))
B = 4
lat = torch.randn(B, 4, 64, 64, device=dev); ctrl = torch.rand(B, 3, 512, 512, device=dev)
emb = torch.randn(3 * B, 81, 768, device=dev, dtype=torch.float16) * 0.1
tt = torch.randint(20, 800, (B,), device=dev)
x3, c3, t3 = torch.cat([lat] * 3), torch.cat([ctrl] * 3), torch.cat([tt] * 3)
for _ in range(12):
    g.forward_unet(x3, c3, t3, emb, True)
torch.cuda.synchronize()
