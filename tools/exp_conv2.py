"""gip_conv3x3_nhwc_f16 vs MIOpen: correctness (against fp32 conv of the same fp16 operands) and time per shape."""
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
def conv(x, w, out):
    N, C, H, W = x.shape
    out.copy_(fused._conv_call(x, w, w.shape[0]))
    return out
shapes = [(2, 64, 64, 9, 7), (12, 320, 320, 64, 64), (12, 640, 320, 64, 64), (12, 960, 320, 64, 64), (12, 640, 640, 32, 32), (12, 1280, 640, 32, 32), (12, 1920, 640, 32, 32),
          (12, 1280, 1280, 16, 16), (12, 2560, 1280, 16, 16), (12, 1280, 1280, 8, 8), (12, 2560, 1280, 8, 8),
          (4, 128, 128, 512, 512), (4, 128, 256, 256, 256), (4, 256, 256, 256, 256), (4, 256, 512, 128, 128), (4, 512, 512, 128, 128), (4, 512, 512, 64, 64)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for N, ci, co, H, W in shapes:
    g = torch.Generator(device=dev).manual_seed(ci + co + H)
    x = torch.randn(N, ci, H, W, device=dev, generator=g).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev, generator=g) * (1.0 / (3 * ci ** 0.5))).half().contiguous(memory_format=torch.channels_last)
    out = torch.empty(N, co, H, W, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last)
    conv(x, w, out)
    ref_h = torch.nn.functional.conv2d(x, w, None, padding=1)
    if N * H * W * co <= 12 * 64 * 64 * 640:
        ref = torch.nn.functional.conv2d(x.float(), w.float(), None, padding=1)
        err, err_m = float((out.float() - ref).abs().max()), float((ref_h.float() - ref).abs().max())
    else:
        err, err_m = float((out.float() - ref_h.float()).abs().max()), float("nan")
    fl = 2.0 * N * H * W * ci * co * 9
    t_m = timed(lambda: torch.nn.functional.conv2d(x, w, None, padding=1))
    t_g = timed(lambda: conv(x, w, out))
    print("N%2d %4d->%4d @%3dx%3d %6.1f GF | max err gip %.2e (miopen %.2e, |ref| %.2f) | miopen %.3f ms %5.0f TF/s | gip %.3f ms %5.0f TF/s  x%.2f" %
          (N, ci, co, H, W, fl / 1e9, err, err_m, float(ref_h.float().abs().max()), t_m, fl / t_m / 1e9, t_g, fl / t_g / 1e9, t_m / t_g), flush=True)
