"""Halo-resident pixel tile (conv3x3_kernel<..., HALO>) against the streaming kernel on the Cin = 128 shapes of the VAE encoder,
same process (debug knob gip_dbg_conv_epilogue = 0 selects the streaming kernel with its per-lane epilogue; GIP_CONV_HALO=0 in
the environment selects the streaming kernel with the LDS epilogue — run the script twice for that comparison)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd import _lib  # noqa: E402
from gaussianip_amd.guidance import fused  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


cl = dict(memory_format=torch.channels_last)
print("GIP_CONV_HALO=%s" % os.environ.get("GIP_CONV_HALO", "1"))
for N, co, H, W, res in [(4, 128, 512, 512, False), (4, 128, 512, 512, True), (4, 256, 256, 256, False), (2, 128, 512, 512, False), (1, 128, 512, 512, False),
                         (1, 128, 1024, 1024, False)]:
    x = torch.randn(N, 128, H, W, device="cuda").half().contiguous(**cl)
    w = (torch.randn(co, 128, 3, 3, device="cuda") / 34.0).half().contiguous(**cl)
    b = torch.randn(co, device="cuda").half()
    r = torch.randn(N, co, H, W, device="cuda").half().contiguous(**cl) if res else None
    t_stats = timed(lambda: fused._conv_call(x, w, co, b, r, []))
    t_plain = timed(lambda: fused._conv_call(x, w, co, b, r, None))
    fl = 2.0 * N * H * W * co * 128 * 9
    print("N%d 128->%3d @%4dx%4d res=%d | with stats %.3f ms %5.0f TFLOP/s | plain %.3f ms %5.0f TFLOP/s" % (
        N, co, H, W, int(res), t_stats, fl / t_stats / 1e9, t_plain, fl / t_plain / 1e9), flush=True)
