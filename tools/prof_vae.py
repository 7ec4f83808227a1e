import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance.networks import VAEEncoder, init_for_benchmark
dev = torch.device("cuda")
vae = init_for_benchmark(VAEEncoder(), 2).to(dev, torch.float16).eval().requires_grad_(False).to(memory_format=torch.channels_last)
img = torch.rand(4, 3, 512, 512, device=dev, requires_grad=True)
for _ in range(12):
    z = vae.encode((img * 2 - 1).half().contiguous(memory_format=torch.channels_last))
    z.sum().backward()
torch.cuda.synchronize()
