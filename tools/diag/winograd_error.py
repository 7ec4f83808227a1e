"""Error and time of the Winograd path against the implicit GEMM on the layer shapes it is used for (fp32 reference).
usage: winograd_error.py"""
import os
import sys

import torch
import torch.nn.functional as F

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


g = torch.Generator(device="cuda").manual_seed(0)
cl = dict(memory_format=torch.channels_last)
for N, cin, cout, res in [(12, 1280, 1280, True), (12, 1280, 1280, False), (12, 2560, 1280, False), (12, 1920, 1280, False)]:
    x = torch.randn(N, cin, 16, 16, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3 * cin ** 0.5)).half().contiguous(**cl)
    b = torch.randn(cout, device="cuda", generator=g).half()
    r = torch.randn(N, cout, 16, 16, device="cuda", generator=g).half().contiguous(**cl) if res else None
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + (0 if r is None else r.float())
    with torch.no_grad():
        os.environ["GIP_WINOGRAD"] = "1"
        ow = fused.conv3x3(x, w, b, r)
        tw = timed(lambda: fused.conv3x3(x, w, b, r))
        os.environ["GIP_WINOGRAD"] = "0"
        od = fused.conv3x3(x, w, b, r)
        td = timed(lambda: fused.conv3x3(x, w, b, r))
    s = float(ref.abs().max())
    print("%4d->%4d res=%d | direct %6.1f us max %.2e rms %.2e | winograd %6.1f us max %.2e rms %.2e  (relative to max|ref| = %.2f)" % (
        cin, cout, res, td, float((od.float() - ref).abs().max()) / s, float((od.float() - ref).pow(2).mean().sqrt()) / s,
        tw, float((ow.float() - ref).abs().max()) / s, float((ow.float() - ref).pow(2).mean().sqrt()) / s, s), flush=True)
