"""Time of gip_conv3x3_nhwc_f16 per layer shape of the AHDS step (no extra copies): TFLOP/s and tile statistics."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused
dev = "cuda"
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
shapes = [(12, 320, 320, 64), (12, 640, 320, 64), (12, 960, 320, 64), (12, 640, 640, 64), (12, 320, 640, 32), (12, 640, 640, 32), (12, 1280, 640, 32),
          (12, 1920, 640, 32), (12, 960, 640, 32), (12, 1280, 1280, 32), (12, 640, 1280, 16), (12, 1280, 1280, 16), (12, 2560, 1280, 16), (12, 1920, 1280, 16),
          (12, 1280, 1280, 8), (12, 2560, 1280, 8),
          (4, 128, 128, 512), (4, 128, 256, 256), (4, 256, 256, 256), (4, 256, 512, 128), (4, 512, 512, 128), (4, 512, 512, 64)]
tot_t = tot_f = 0.0
for N, ci, co, H in shapes:
    x = torch.randn(N, ci, H, H, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev) * (1.0 / (3 * ci ** 0.5))).half().contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * H * ci * co * 9
    t = timed(lambda: fused._conv_call(x, w, co))
    bn = 160 if (co % 160 == 0 and co % 128 != 0) else 128
    tiles = ((N * H * H + 127) // 128) * ((co + bn - 1) // bn)
    print("N%2d %4d->%4d @%3d  %6.1f GF  %.3f ms  %5.0f TF/s   tiles %5d (%.2f rounds of 512)" % (N, ci, co, H, fl / 1e9, t, fl / t / 1e9, tiles, tiles / 512.0), flush=True)
