"""256-row 8-wave convolution kernel (conv_big_kernel) against the 128-row kernel: correctness (output, residual path,
epilogue statistics) and interleaved timing in one process."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd import _lib  # noqa: E402
from gaussianip_amd.guidance import fused  # noqa: E402

lib = _lib.nn_lib()
BIG = ctypes.c_int.in_dll(lib._lib, "gip_dbg_conv_big")

dev = "cuda"
shapes = [(4, 128, 128, 512, 512), (4, 128, 256, 256, 256), (4, 256, 256, 256, 256), (4, 256, 512, 128, 128), (4, 512, 512, 128, 128),
          (12, 640, 640, 32, 32), (12, 1280, 640, 32, 32), (12, 1920, 640, 32, 32), (4, 512, 512, 64, 64), (12, 320, 640, 32, 32)]
# round 6: the 384 x 160 tile (Cout = 320 at 12 x 64^2: 256 tiles) against the 128 x 160 tile (768 workgroups)
shapes_320 = [(12, 320, 320, 64, 64), (12, 640, 320, 64, 64), (12, 960, 320, 64, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "320":
    shapes = shapes_320
elif len(sys.argv) > 1 and sys.argv[1] == "512":      # the 128 x 256 tile (VAE 512 channels at 4 x 64^2): run with GIP_CONV_384=3
    shapes = [(4, 512, 512, 64, 64)]
elif len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for N, ci, co, H, W in shapes:
    g = torch.Generator(device=dev).manual_seed(ci + co + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, ci, H, W, device=dev, generator=g).half().contiguous(**cl)
    w = (torch.randn(co, ci, 3, 3, device=dev, generator=g) / (3 * ci ** 0.5)).half().contiguous(**cl)
    b = torch.randn(co, device=dev, generator=g).half()
    res = torch.randn(N, co, H, W, device=dev, generator=g).half().contiguous(**cl)
    fl = 2.0 * N * H * W * ci * co * 9

    def run(big, r, st=None, stg=1):
        BIG.value = big
        y = fused._conv_call(x, w, co, b, res if r else None, st)
        BIG.value = -1
        return y

    y0, y1 = run(0, False), run(1, False)
    e_plain = float((y0.float() - y1.float()).abs().max())
    y0r, y1r = run(0, True), run(1, True)
    e_res = float((y0r.float() - y1r.float()).abs().max())
    h0, h1 = [], []
    run(0, True, h0)
    ys = run(1, True, h1)
    e_st = float((h0[0] - h1[0]).abs().max() / h0[0].abs().max()) if h0 and h1 else float("nan")
    e_st_out = float((ys.float() - y1r.float()).abs().max())
    res_t = {}
    e_nostg = float((run(1, True, None, 0).float() - y1r.float()).abs().max())
    fns = {"old": lambda: run(0, False), "big": lambda: run(1, False, None, 0), "big+stagger": lambda: run(1, False, None, 1),
           "old+res": lambda: run(0, True), "big+stg+res": lambda: run(1, True)}
    for rnd in range(5):
        for k, f in fns.items():
            f()
            f()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                f()
            e.record()
            torch.cuda.synchronize()
            res_t.setdefault(k, []).append(a.elapsed_time(e) / 10)
    print("N%2d %4d->%4d @%3dx%3d %6.1f GF | " % (N, ci, co, H, W, fl / 1e9) +
          " | ".join("%s %.1f us %4.0f TF" % (k, sorted(v)[2] * 1e3, fl / sorted(v)[2] / 1e9) for k, v in res_t.items()) +
          " | maxdiff plain %.2e res %.2e nostagger %.2e stats(rel) %.2e stats-out %.2e |y| %.2f" % (e_plain, e_res, e_nostg, e_st, e_st_out, float(y0.float().abs().max())), flush=True)
