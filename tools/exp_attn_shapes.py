"""Self-attention kernel timing at the denoiser shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for B, H, N, D in [(12, 8, 4096, 40), (4, 8, 4096, 40), (12, 8, 1024, 80), (6, 8, 4096, 40), (3, 8, 4096, 40)]:
    q, k, v = [torch.randn(B, N, H * D, device="cuda").half() for _ in range(3)]
    with torch.no_grad():
        t = min(timed(lambda: fused.attention(q, k, v, H)) for _ in range(3))
    fl = 4.0 * B * H * N * N * D
    print("B %2d H %d N %4d D %3d: %.4f ms  %5.0f TFLOP/s" % (B, H, N, D, t, fl / t / 1e9), flush=True)
