"""Same-process A/B of the VAE encoder fwd+bwd: dilated-gradient downsample on / off."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused
from gaussianip_amd.guidance.networks import VAEEncoder, init_for_benchmark
import torch.nn.functional as F
dev = torch.device("cuda")
vae = init_for_benchmark(VAEEncoder(), 2).to(dev, torch.float16).eval().requires_grad_(False).to(memory_format=torch.channels_last)
img = torch.rand(4, 3, 512, 512, device=dev, requires_grad=True)
def run():
    z = vae.encode((img * 2 - 1).half().contiguous(memory_format=torch.channels_last)); z.sum().backward()
def wall(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
orig = fused.downsample_asym
plain = lambda x, w, b: F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
for _ in range(4): run()
fused.downsample_asym = plain
for _ in range(4): run()
for rep in range(3):
    fused.downsample_asym = orig; a = wall()
    fused.downsample_asym = plain; b = wall()
    print("dilated dgrad %.2f ms | library dgrad %.2f ms" % (a, b), flush=True)
