"""torch.profiler kernel table of the stage-3 step's LPIPS part (forward to cached target features + backward)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance.perceptual import LPIPSVGG
dev = torch.device("cuda")
lp = LPIPSVGG().init_for_benchmark(0).prepare_inference(dev)
a = torch.rand(4, 3, 415, 290, device=dev); b = torch.rand(4, 3, 415, 290, device=dev)
tf = lp.target_features(b, True)
def step():
    x = a.clone().requires_grad_(True)
    lp.distance_to_features(x, tf, True).mean().backward()
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("LPIPS fwd+bwd %.2f ms" % ((time.perf_counter() - t0) * 100))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=70))
