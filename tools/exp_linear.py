import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch.nn.functional as F
from gaussianip_amd.guidance import fused
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
torch.manual_seed(0)
with torch.no_grad():
    for M, K, N in [(49152, 320, 320), (49152, 320, 960), (12288, 640, 640), (3072, 1280, 1280), (49152, 1280, 320), (12288, 2560, 640), (3072, 5120, 1280)]:
        x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).half(); b = torch.randn(N, device="cuda").half(); r = torch.randn(M, N, device="cuda").half()
        ref = F.linear(x.float(), w.float(), b.float()) + r.float()
        got = fused.linear(x, w, b, r)
        err = float((got.float() - ref).abs().max()) / float(ref.abs().max())
        t_t = timed(lambda: F.linear(x, w, b) + r); t_g = timed(lambda: fused.linear(x, w, b, r))
        print("linear+res M%6d K%5d N%5d | rel err %.1e | torch %.3f ms | gip %.3f ms x%.2f" % (M, K, N, err, t_t, t_g, t_t / t_g), flush=True)
    for M, K, D in [(49152, 320, 1280), (12288, 640, 2560), (3072, 1280, 5120)]:
        x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(2 * D, K, device="cuda") / K ** 0.5).half(); b = torch.randn(2 * D, device="cuda").half()
        y = F.linear(x.float(), w.float(), b.float()); v, g = y.chunk(2, -1); ref = v * F.gelu(g)
        got = fused.linear(x, w, b, None, True)
        err = float((got.float() - ref).abs().max()) / float(ref.abs().max())
        t_t = timed(lambda: fused.geglu(F.linear(x, w, b))); t_g = timed(lambda: fused.linear(x, w, b, None, True))
        print("geglu      M%6d K%5d D%5d | rel err %.1e | hipblaslt+geglu %.3f ms | gip %.3f ms x%.2f" % (M, K, D, err, t_t, t_g, t_t / t_g), flush=True)
