"""Fused GroupNorm kernels: achieved bandwidth per shape (forward = 3 passes over the tensor, backward = 5)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance.fused import GroupNormAct
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for N, C, H in [(12, 320, 64), (12, 640, 64), (12, 960, 64), (12, 640, 32), (12, 1280, 32), (12, 1280, 16), (12, 2560, 16), (12, 1280, 8), (4, 128, 512), (4, 128, 256), (4, 256, 256), (4, 256, 128), (4, 512, 128), (4, 512, 64)]:
    x = torch.randn(N, C, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    m = GroupNormAct(32, C, act=True).cuda().half().requires_grad_(False)
    y = m(x)
    dy = torch.randn_like(y)
    tf = timed(lambda: m(x))
    tb = timed(lambda: torch.autograd.grad(m(x), x, dy)) - tf
    nb = x.numel() * 2
    print("N%2d C%4d @%3d  %6.1f MB | fwd %.3f ms %5.0f GB/s | bwd %.3f ms %5.0f GB/s" % (N, C, H, nb / 1e6, tf, 3 * nb / tf / 1e6, tb, 5 * nb / tb / 1e6), flush=True)
