"""Would Winograd F(2x2, 3x3) pay at the 16 x 16 level of the U-Net?  Times the sixteen batched GEMMs its middle stage needs
([16, M / 4, Cin] x [16, Cin, Cout]) against the implicit-GEMM convolution that does the whole layer now — the transforms
would come on top.  usage: winograd_feasibility.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return a.elapsed_time(b) / n * 1e3


for N, H, Cin, Cout in [(12, 16, 1280, 1280), (12, 16, 2560, 1280), (12, 32, 640, 640), (12, 32, 1280, 640), (12, 8, 1280, 1280)]:
    x = torch.randn(N, Cin, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.01).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t_conv = timed(lambda: fused.conv3x3(x, w))
        M4 = N * H * H // 4
        a = torch.randn(16, M4, Cin, device="cuda").half()
        b = torch.randn(16, Cout, Cin, device="cuda").half()
        t_bmm = timed(lambda: torch.bmm(a, b.transpose(1, 2)))
        # the transforms move: input 1x read + 4x write, output 4x read + 1x write (+ residual): elementwise stand-ins of that traffic
        xin = torch.randn(M4 * 4, Cin, device="cuda").half()
        t_in = timed(lambda: a.copy_(xin.view(4, M4, Cin).repeat(4, 1, 1)))
        t_out = timed(lambda: a[:4].sum(0))
    print("N%d %dx%d %d->%d | conv now %6.1f us | 16 GEMMs %6.1f us | transform stand-ins %5.1f + %5.1f us" %
          (N, H, H, Cin, Cout, t_conv, t_bmm, t_in, t_out), flush=True)
