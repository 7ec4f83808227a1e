"""A/B of the conv3x3 kernel's round-3 knobs in ONE process (interleaved rounds, median): tile order (m-major / n-major
inside an XCD's chunk), epilogue (per-lane 8-byte stores / coalesced through LDS), split-K factor.  Correctness of every
variant against the fp32 convolution of the same operands.  usage: exp_conv5.py [unet|vae|all]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd import _lib  # noqa: E402
from gaussianip_amd.guidance import fused  # noqa: E402

lib = _lib.nn_lib()
dev = "cuda"


def knob(name):
    return ctypes.c_int.in_dll(lib._lib, name)


ORDER, EPI, KSPLIT = knob("gip_dbg_conv_order"), knob("gip_dbg_conv_epilogue"), knob("gip_dbg_conv_ksplit")


def time_variants(fns, rounds=7, inner=10):
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                f()
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / inner)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


UNET = [(12, 320, 320, 64, 64), (12, 640, 640, 32, 32), (12, 1280, 640, 32, 32), (12, 1920, 640, 32, 32), (12, 1280, 1280, 16, 16),
        (12, 2560, 1280, 16, 16), (12, 1280, 1280, 8, 8), (12, 2560, 1280, 8, 8), (12, 960, 320, 64, 64)]
VAE = [(4, 128, 128, 512, 512), (4, 256, 256, 256, 256), (4, 512, 512, 128, 128), (4, 512, 512, 64, 64)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
shapes = {"unet": UNET, "vae": VAE, "all": UNET + VAE}[which]
for N, ci, co, H, W in shapes:
    g = torch.Generator(device=dev).manual_seed(ci + co + H)
    x = torch.randn(N, ci, H, W, device=dev, generator=g).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev, generator=g) * (1.0 / (3 * ci ** 0.5))).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=dev, generator=g).half()
    res = torch.randn(N, co, H, W, device=dev, generator=g).half().contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * W * ci * co * 9
    small = N * H * W * co <= 12 * 64 * 64 * 640
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1) if small else None

    def run(order, epi, ks, r):
        ORDER.value, EPI.value, KSPLIT.value = order, epi, ks
        y = fused._conv_call(x, w, co, b, res if r else None)
        ORDER.value, EPI.value, KSPLIT.value = -1, -1, 0
        return y

    base = run(0, 0, 0, False)
    errs = {}
    for name, (o, e, ks) in {"old": (0, 0, 0), "lds": (0, 1, 0), "nmaj+lds": (1, 1, 0)}.items():
        y = run(o, e, ks, False)
        errs[name] = float((y.float() - (ref if small else base.float())).abs().max())
        yr = run(o, e, ks, True)
        want = (ref + res.float()) if small else (base.float() + res.float())
        errs[name + "+res"] = float((yr.float() - want).abs().max())
    fns = {}
    for r in (False, True):
        tag = "+res" if r else ""
        fns["old" + tag] = lambda r=r: run(0, 0, 0, r)
        fns["lds" + tag] = lambda r=r: run(0, 1, 0, r)
        fns["nmaj+lds" + tag] = lambda r=r: run(1, 1, 0, r)
    tiles = fused._conv_tiles(N, H, W, co)
    if tiles < 256:
        for ks in (2, 4, 6, 9):
            fns["lds ks%d" % ks] = lambda ks=ks: run(0, 1, ks, False)
            fns["nmaj ks%d" % ks] = lambda ks=ks: run(1, 1, ks, False)
    t = time_variants(fns)
    print("N%2d %4d->%4d @%3dx%3d %6.1f GF tiles %5d | " % (N, ci, co, H, W, fl / 1e9, tiles) +
          " | ".join("%s %.1f us %4.0f TF" % (k, v * 1e3, fl / v / 1e9) for k, v in t.items()), flush=True)
    print("      max err: " + ", ".join("%s %.2e" % kv for kv in errs.items()), flush=True)
