import ctypes, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd import _lib
from gaussianip_amd.guidance import fused
lib = _lib.nn_lib()
dev = "cuda"
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for N, ci, co, H, W in [(4, 512, 512, 64, 64), (8, 512, 512, 64, 64), (16, 512, 512, 64, 64), (4, 512, 512, 96, 96), (4, 512, 512, 128, 128), (1, 512, 512, 128, 128),
                        (4, 576, 512, 64, 64), (4, 512, 640, 64, 64), (4, 448, 512, 64, 64), (4, 256, 256, 64, 64), (4, 1024, 512, 64, 64), (3, 512, 512, 64, 64), (5, 512, 512, 64, 64)]:
    x = torch.randn(N, ci, H, W, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev) * 0.01).half().contiguous(memory_format=torch.channels_last)
    out = torch.empty(N, co, H, W, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last)
    def conv():
        fused._conv_call(x, w, co)
    fl = 2.0 * N * H * W * ci * co * 9
    t = timed(conv)
    bn = 160 if (co % 160 == 0 and co % 128) else 128
    print("N%2d %4d->%4d @%3d  blocks %5d  %.3f ms %5.0f TF/s" % (N, ci, co, H, (N * H * W + 127) // 128 * ((co + bn - 1) // bn), t, fl / t / 1e9), flush=True)
