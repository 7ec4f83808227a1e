"""The Winograd path at the batch sizes of a sharded step (configs[3]: 1 or 2 views per rank = batch 3 or 6 in the denoise).
usage: winograd_small_batch.py"""
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


g = torch.Generator(device="cuda").manual_seed(0)
cl = dict(memory_format=torch.channels_last)
for N in (3, 6):
    for H, cin, cout in [(16, 640, 1280), (16, 1280, 1280), (16, 2560, 1280), (32, 960, 640), (32, 1280, 640), (32, 1920, 640)]:
        x = torch.randn(N, cin, H, H, device="cuda", generator=g).half().contiguous(**cl)
        w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3 * cin ** 0.5)).half().contiguous(**cl)
        b = torch.randn(cout, device="cuda", generator=g).half()
        with torch.no_grad():
            os.environ["GIP_WINOGRAD_SHAPES"] = "%d:%d" % (H, cin)
            tw = timed(lambda: fused.conv3x3(x, w, b, None, gn_next=True))
            os.environ["GIP_WINOGRAD_SHAPES"] = ""
            td = timed(lambda: fused.conv3x3(x, w, b, None, gn_next=True))
        print("N=%d %2dx%-2d %4d->%4d | implicit GEMM %6.1f us | winograd %6.1f us  %s" % (N, H, H, cin, cout, td, tw, "<-- gains" if tw < 0.95 * td else ""), flush=True)
