import sys, os, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, fused
from gaussianip_amd.guidance.ahds import AHDSSchedule
dev = torch.device("cuda")
g = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
B = 4
lat = torch.randn(3 * B, 4, 64, 64, device=dev); ctrl = torch.rand(B, 3, 512, 512, device=dev)
emb = torch.randn(3 * B, 81, 768, device=dev, dtype=torch.float16) * 0.1; tt = torch.randint(20, 800, (3 * B,), device=dev)
def run():
    with torch.no_grad():
        return g.forward_unet(lat, ctrl, tt, emb, True)
def wall(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
orig = fused.downsample_sym
lib = lambda x, w, b: F.conv2d(x, w, b, stride=2, padding=1)
for _ in range(4): run()
fused.downsample_sym = lib
for _ in range(4): run()
for rep in range(3):
    fused.downsample_sym = orig; a = wall()
    fused.downsample_sym = lib; b = wall()
    print("stride-2 on the MFMA kernel %.2f ms | MIOpen %.2f ms" % (a, b), flush=True)
