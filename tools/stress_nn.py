"""Race screen for the LDS-DMA convolution and the attention kernel: many repetitions at several shapes, every result
compared bitwise with the first one (the kernels are deterministic by construction) while other streams keep the GPU busy."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
cl = dict(memory_format=torch.channels_last)
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device=dev).half()
bad = 0
for (N, ci, co, H) in [(12, 320, 320, 64), (12, 640, 1280, 16), (4, 128, 128, 256), (12, 1280, 1280, 8), (3, 192, 72, 33)]:
    x = torch.randn(N, ci, H, H, device=dev, generator=g).half().contiguous(**cl)
    w = (torch.randn(co, ci, 3, 3, device=dev, generator=g) / (3 * ci ** 0.5)).half().contiguous(**cl)
    first = fused._conv_call(x, w, co).clone()
    for it in range(200):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise_a @ noise_a          # concurrent load on another stream
        out = fused._conv_call(x, w, co)
        if not torch.equal(out, first):
            bad += 1
    print("conv", (N, ci, co, H), "mismatching repeats:", bad, flush=True)
for (B, Hh, Nq, Nkv, D) in [(12, 8, 4096, 4096, 40), (2, 8, 1024, 77, 40), (2, 8, 1024, 2048, 80), (2, 3, 512, 512, 64)]:
    q = torch.randn(B, Nq, Hh * D, device=dev, generator=g).half()
    k = torch.randn(B, Nkv, Hh * D, device=dev, generator=g).half(); v = torch.randn(B, Nkv, Hh * D, device=dev, generator=g).half()
    with torch.no_grad():
        first = fused.attention(q, k, v, Hh).clone()
        for it in range(100):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    noise_a @ noise_a
            if not torch.equal(fused.attention(q, k, v, Hh), first):
                bad += 1
    print("attn", (B, Hh, Nq, Nkv, D), "mismatching repeats:", bad, flush=True)
torch.cuda.synchronize()
print("TOTAL MISMATCHES", bad)
