"""Every dense GEMM shape of the denoise (batch 12 and the 6 / 3 of sharded steps): hipBLASLt (F.linear) against the MFMA linear
of this repo (fused.linear = conv3x3_kernel<TAPS = 1>), same process, alternating.  Prints one line per shape; the dispatch in
guidance/networks.py keeps the library only where it is >= 5 % faster.  JSON lines at the end."""
import json
import os
import sys

import torch
import torch.nn.functional as F

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


rows = []
for batch in (12, 6, 3):
    for hw, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
        M = batch * hw
        shapes = [("proj_in / to_q / zero conv", M, C, C, True, False), ("qkv", M, C, 3 * C, False, False),
                  ("to_out (+residual)", M, C, C, True, True), ("ff_in (8C)", M, C, 8 * C, True, False), ("ff_out (+residual)", M, 4 * C, C, True, True)]
        if hw == 4096:
            shapes.append(("shortcut 960->320", M, 960, 320, True, False))
            shapes.append(("shortcut 640->320", M, 640, 320, True, False))
        if hw == 1024:
            shapes.append(("shortcut 320->640", M, 320, 640, True, False))
            shapes.append(("shortcut 1920->640", M, 1920, 640, True, False))
        if hw == 256:
            shapes.append(("shortcut 640->1280", M, 640, 1280, True, False))
            shapes.append(("shortcut 2560->1280", M, 2560, 1280, True, False))
        for name, m, k, n, has_bias, has_res in shapes:
            x = torch.randn(m, k, device="cuda").half()
            w = (torch.randn(n, k, device="cuda") / k ** 0.5).half()
            b = torch.randn(n, device="cuda").half() if has_bias else None
            r = torch.randn(m, n, device="cuda").half() if has_res else None
            with torch.no_grad():
                lib = (lambda: F.linear(x, w, b) + r) if has_res else (lambda: F.linear(x, w, b))
                own = lambda: fused.linear(x, w, b, r)  # noqa: E731
                assert fused.linear_supported(x, w)
                t_lib = min(timed(lib), timed(lib))
                t_own = min(timed(own), timed(own))
                err = float((own().float() - lib().float()).abs().max())
            fl = 2.0 * m * k * n
            rows.append(dict(batch=batch, tokens=hw, name=name, M=m, K=k, N=n, residual=has_res, lib_us=round(t_lib * 1e3, 1), own_us=round(t_own * 1e3, 1),
                             lib_tflops=round(fl / t_lib / 1e9, 0), own_tflops=round(fl / t_own / 1e9, 0), own_over_lib=round(t_own / t_lib, 3)))
            print("b%2d %4d tok %-26s M %6d K %5d N %5d | hipBLASLt %7.1f us %5.0f TF | own %7.1f us %5.0f TF | own/lib %.2f  maxdiff %.3g" % (
                batch, hw, name, m, k, n, t_lib * 1e3, fl / t_lib / 1e9, t_own * 1e3, fl / t_own / 1e9, t_own / t_lib, err), flush=True)
for r_ in rows:
    print(json.dumps(r_))
