import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch.nn.functional as F
from gaussianip_amd.guidance import fused
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B, H, N, D in [(12, 8, 4096, 40), (12, 8, 1024, 40), (12, 8, 4096, 64), (4, 8, 4096, 40), (12, 8, 1024, 80), (2, 8, 4096, 80)]:
    q, k, v = [torch.randn(B, N, H * D, device="cuda").half() for _ in range(3)]
    sp = lambda t: t.view(B, N, H, D).transpose(1, 2)
    with torch.no_grad():
        t_s = timed(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, H * D))
        t_g = timed(lambda: fused.attention(q, k, v, H))
    fl = 4.0 * B * H * N * N * D
    print("B%2d H%d N%4d D%2d | sdpa %.3f ms %4.0f TF/s | gip %.3f ms %4.0f TF/s  x%.2f" % (B, H, N, D, t_s, fl / t_s / 1e9, t_g, fl / t_g / 1e9, t_s / t_g), flush=True)
