"""An attention kernel variant selected by environment switches (read once per process) against the plain loop: bit-equality + timing.
Run once per setting; GIP_ATTN_REF=1 writes the reference outputs to /tmp first (use GIP_ATTN_NW=4 for that run)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


ref_run = os.environ.get("GIP_ATTN_REF") == "1"
print({k: v for k, v in os.environ.items() if k.startswith("GIP_ATTN")})
torch.manual_seed(0)
for B, H, N, D in [(12, 8, 4096, 40), (8, 8, 4096, 40), (4, 8, 4096, 40), (3, 8, 4096, 40), (2, 8, 16384, 40), (1, 8, 4096 + 128, 40)]:
    q, k, v = [torch.randn(B, N, H * D, device="cuda").half() for _ in range(3)]
    ragged = N == 4096 and B == 3
    kk = k[:, : N - 37].contiguous() if ragged else k
    vv = v[:, : N - 37].contiguous() if ragged else v
    with torch.no_grad():
        out = fused.attention(q, kk, vv, H)
        t = min(timed(lambda: fused.attention(q, kk, vv, H)) for _ in range(3))
    path = "/tmp/attn_ref_%d_%d_%d_%d.pt" % (B, H, N, D)
    cmp = ""
    if ref_run:
        torch.save(out.cpu(), path)
    else:
        ref = torch.load(path).cuda()
        cmp = "  equal %s (max|diff| %.3e)" % (bool(torch.equal(out, ref)), (out.float() - ref.float()).abs().max().item())
    fl = 4.0 * B * H * N * kk.shape[1] * D
    print("B %2d H %d N %5d D %3d: %.4f ms  %5.0f TFLOP/s%s" % (B, H, N, D, t, fl / t / 1e9, cmp), flush=True)
