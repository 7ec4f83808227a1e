"""VAE encoder forward + backward (4 x 512^2, eager launches), wall time per call; environment switches (GIP_CONV_HALO, ...) are read
at start-up: run it alternately under both settings on one box."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance.networks import VAEEncoder, init_for_benchmark  # noqa: E402

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
vae = init_for_benchmark(VAEEncoder(), 2).to(dev, torch.float16).eval().requires_grad_(False).to(memory_format=torch.channels_last)
img = torch.rand(B, 3, 512, 512, device=dev, requires_grad=True)


def run():
    z = vae.encode((img * 2 - 1).half().contiguous(memory_format=torch.channels_last))
    z.sum().backward()


for _ in range(5):
    run()
res = []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 10 * 1e3)
print("VAE encoder fwd+bwd B=%d  %s  ms per call: %s" % (B, " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("GIP_")),
                                                       " ".join("%.3f" % r for r in res)), flush=True)
