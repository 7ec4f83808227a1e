"""Timing ablations of the conv3x3 main loop (results are WRONG on purpose): which of {activation DMA, weight DMA, LDS reads
+ MFMA} the K loop is waiting for.  bit0: no A (activation) DMA, bit1: no B (weight) DMA, bit2: no LDS reads / MFMA."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd import _lib
from gaussianip_amd.guidance import fused
lib = _lib.nn_lib()
AB = ctypes.c_int.in_dll(lib._lib, "gip_dbg_conv_ablate")
dev = "cuda"
shapes = [(12, 320, 320, 64, 64), (12, 1280, 640, 32, 32), (4, 128, 128, 512, 512), (4, 256, 256, 256, 256), (4, 512, 512, 128, 128)]
names = {0: "full", 1: "no A dma", 2: "no B dma", 3: "no dma", 4: "dma only (no mfma)", 7: "barriers only"}
for N, ci, co, H, W in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, ci, H, W, device=dev, generator=g).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev, generator=g) / (3 * ci ** 0.5)).half().contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * W * ci * co * 9
    res = {}
    for rnd in range(5):
        for ab in names:
            AB.value = ab
            for _ in range(2): fused._conv_call(x, w, co)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fused._conv_call(x, w, co)
            b.record(); torch.cuda.synchronize()
            res.setdefault(ab, []).append(a.elapsed_time(b) / 10)
    AB.value = 0
    print("N%2d %4d->%4d @%3dx%3d | " % (N, ci, co, H, W) + " | ".join("%s %.1f us (%.0f TF-equiv)" % (names[k], sorted(v)[2] * 1e3, fl / sorted(v)[2] / 1e9) for k, v in res.items()), flush=True)
