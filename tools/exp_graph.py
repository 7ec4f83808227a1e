"""Does capturing the ControlNet + U-Net denoise in a HIP graph pay?  eager vs graph replay, batch 12 at 64^2 latents."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
from gaussianip_amd.guidance.ahds import AHDSSchedule
dev = torch.device("cuda")
g = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
B = 4
lat = torch.randn(3 * B, 4, 64, 64, device=dev)
ctrl = torch.rand(B, 3, 512, 512, device=dev)
emb = torch.randn(3 * B, 81, 768, device=dev, dtype=torch.float16) * 0.1
tt = torch.randint(20, 800, (3 * B,), device=dev)
def run():
    with torch.no_grad():
        return g.forward_unet(lat, ctrl, tt, emb, True, replicas=3)
for _ in range(4):
    ref = run()
torch.cuda.synchronize()
def wall(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, h / n * 1e3
print("eager  %.2f ms (host enqueue %.2f ms)" % wall(run))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): run()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = run()
graph.replay(); torch.cuda.synchronize()
print("graph  %.2f ms (host enqueue %.2f ms)" % wall(graph.replay), " max diff vs eager", float((out - ref).abs().max()))
