"""GroupNorm apply passes against a plain copy of the same tensors (what does the memory system give a one-read-one-write
stream at these sizes, and how close are the apply kernels?).  Times per pass and GB/s of the bytes each pass must move."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused  # noqa: E402
from gaussianip_amd.guidance.fused import GroupNormAct  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


cl = dict(memory_format=torch.channels_last)
for N, C, H in [(4, 128, 512), (4, 128, 256), (4, 256, 256), (4, 256, 128), (4, 512, 128), (4, 512, 64), (12, 320, 64), (12, 640, 64), (12, 960, 64),
                (12, 640, 32), (12, 1280, 32), (12, 1280, 16), (12, 2560, 16)]:
    x = torch.randn(N, C, H, H, device="cuda").half().contiguous(**cl)
    dy = torch.randn_like(x)
    acc = torch.randn_like(x)
    out = torch.empty_like(x)
    gn = GroupNormAct(32, C, act=True).cuda().half().requires_grad_(False)
    nb = x.numel() * 2
    t_copy = timed(lambda: out.copy_(x))
    y, mean, rstd = fused._gn_fwd_raw(x, gn, None, None)
    t_fwd_full = timed(lambda: fused._gn_fwd_raw(x, gn, None, None))                 # statistics pass + apply
    st = None
    if (H * H) % 128 == 0:
        st = torch.randn(N * H * H // 128, C, 2, device="cuda").abs()
        t_fwd_apply = timed(lambda: fused._gn_fwd_raw(x, gn, None, st))             # finalize + apply only
    else:
        t_fwd_apply = float("nan")
    t_bwd = timed(lambda: fused._gn_bwd_raw(x, dy, gn, mean, rstd, None))             # reduce (2 reads) + apply (2 reads, 1 write)
    t_bwd_acc = timed(lambda: fused._gn_bwd_raw(x, dy, gn, mean, rstd, None, accum=acc))
    sums = torch.randn(N * H * H // 128, C, 2, device="cuda") if (H * H) % 128 == 0 else None
    t_bwd_sums = timed(lambda: fused._gn_bwd_raw(x, dy, gn, mean, rstd, None, accum=acc, chan_sums=sums)) if sums is not None else float("nan")
    print("N%2d C%4d @%3d %6.1f MB | copy %.3f ms %5.0f GB/s | fwd stats+apply %.3f (%5.0f) apply-only %.3f (%5.0f GB/s of 2T) | "
          "bwd reduce+apply %.3f (%5.0f of 5T) +accum %.3f (%5.0f of 6T) apply-only+accum %.3f (%5.0f of 4T)" % (
              N, C, H, nb / 1e6, t_copy, 2 * nb / t_copy / 1e6, t_fwd_full, 3 * nb / t_fwd_full / 1e6, t_fwd_apply, 2 * nb / t_fwd_apply / 1e6,
              t_bwd, 5 * nb / t_bwd / 1e6, t_bwd_acc, 6 * nb / t_bwd_acc / 1e6, t_bwd_sums, 4 * nb / t_bwd_sums / 1e6), flush=True)

print("LayerNorm (one read, one write):")
for M, C in [(49152, 320), (16384, 320), (12288, 640), (3072, 1280), (768, 1280), (24576, 320), (6144, 640), (1536, 1280)]:
    x = torch.randn(M, C, device="cuda").half()
    ln = fused.LayerNorm(C).cuda().half().requires_grad_(False)
    out = torch.empty_like(x)
    with torch.no_grad():
        t_ln = timed(lambda: ln(x), 50)
    t_cp = timed(lambda: out.copy_(x), 50)
    nb = x.numel() * 2
    print("M %6d C %4d %5.1f MB | layernorm %.4f ms %5.0f GB/s | copy %.4f ms %5.0f GB/s" % (M, C, nb / 1e6, t_ln, 2 * nb / t_ln / 1e6, t_cp, 2 * nb / t_cp / 1e6), flush=True)
