"""The decoupled cross-attention (77 text + 4 image-prompt keys, own softmax each) at the 64 x 64 / 32 x 32 levels under the
GIP_ATTN_* switches: bit-equality against the reference run's outputs + timing.  GIP_ATTN_REF=1 writes the reference."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused  # noqa: E402


def timed(fn, n=50):
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


ref_run = os.environ.get("GIP_ATTN_REF") == "1"
print({k: v for k, v in os.environ.items() if k.startswith("GIP_ATTN")})
torch.manual_seed(0)
for B, H, N, D in [(12, 8, 4096, 40), (4, 8, 4096, 40), (12, 8, 1024, 80)]:
    q = torch.randn(B, N, H * D, device="cuda").half()
    k, v = [torch.randn(B, 77, H * D, device="cuda").half() for _ in range(2)]
    k2, v2 = [torch.randn(B, 4, H * D, device="cuda").half() for _ in range(2)]
    with torch.no_grad():
        out = fused.attention(q, k, v, H, k2, v2, 0.7)
        t = min(timed(lambda: fused.attention(q, k, v, H, k2, v2, 0.7)) for _ in range(3))
    path = "/tmp/attn_cross_ref_%d_%d_%d_%d.pt" % (B, H, N, D)
    cmp = ""
    if ref_run:
        torch.save(out.cpu(), path)
    else:
        ref = torch.load(path).cuda()
        cmp = "  equal %s" % bool(torch.equal(out, ref))
    print("B %2d H %d N %5d D %3d: %.4f ms%s" % (B, H, N, D, t, cmp), flush=True)
