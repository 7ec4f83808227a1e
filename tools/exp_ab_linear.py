"""Same-process A/B of the denoise with / without the MFMA linear epilogue fusions."""
import sys, os, time, torch
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
orig = fused.linear_supported
for _ in range(4): run()
fused.linear_supported = lambda *a: False
for _ in range(4): run()
fused.linear_supported = orig
for rep in range(3):
    a = wall()
    fused.linear_supported = lambda *a: False
    b = wall()
    fused.linear_supported = orig
    print("fused linear %.2f ms | hipBLASLt path %.2f ms" % (a, b), flush=True)
