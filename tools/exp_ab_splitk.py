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
for _ in range(4): run()
fused._MIN_CONV_TILES = 256
for _ in range(4): run()
for rep in range(3):
    fused._MIN_CONV_TILES = 32; a = wall()
    fused._MIN_CONV_TILES = 256; b = wall()
    print("split-K small convs %.2f ms | MIOpen for < 256 tiles %.2f ms" % (a, b), flush=True)
