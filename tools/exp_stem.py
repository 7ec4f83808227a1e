"""ControlNet hint stem (3->16->16->32s2->32->96s2->96->256s2->320) at 1024^2: per-convolution time by layout / path."""
import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dev = torch.device("cuda")
specs = [(3, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1), (96, 256, 2), (256, 320, 1)]
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = int(os.environ.get("RES", 1024))
H = res
for cin, cout, s in specs:
    x = torch.randn(1, cin, H, H, device=dev).half()
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).half(); b = torch.zeros(cout, device=dev).half()
    xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    a = t(lambda: F.conv2d(xl, wl, b, stride=s, padding=1))
    c = t(lambda: F.conv2d(x, w, b, stride=s, padding=1))
    d = t(lambda: F.conv2d(x.float(), w.float(), b.float(), stride=s, padding=1))
    print("%3d -> %3d s%d @%4d: NHWC fp16 %7.3f ms | NCHW fp16 %7.3f ms | NCHW fp32 %7.3f ms" % (cin, cout, s, H, a, c, d), flush=True)
    H //= s
