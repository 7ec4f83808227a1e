"""Race screen for the LDS-DMA convolution, the attention kernel and (round 5) the narrow-tile GEMM, the GroupNorm-in-convolution /
-in-Winograd fusions, the row softmax and the single-accumulator decoupled cross-attention: many repetitions at several shapes, every result
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


def repeat(name, fn, n=100):
    """fn() -> tensor or tuple of tensors; every repetition must equal the first bit for bit"""
    global bad
    with torch.no_grad():
        first = fn()
        first = [t.clone() for t in (first if isinstance(first, (tuple, list)) else (first,))]
        miss = 0
        for it in range(n):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    noise_a @ noise_a
            out = fn()
            out = out if isinstance(out, (tuple, list)) else (out,)
            if not all(torch.equal(a, b) for a, b in zip(out, first)):
                miss += 1
    bad += miss
    print(name, "mismatching repeats:", miss, flush=True)


# ---- round 5 kernels ----
import ctypes  # noqa: E402
from gaussianip_amd import _lib  # noqa: E402
lib = _lib.nn_lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
# own GEMM on 128 x 64 tiles (small grids), plain / row sums / GroupNorm statistics
for (M, K, Nn) in [(768, 1280, 1280), (3072, 320, 960), (192, 640, 640)]:
    x = torch.randn(M, K, device=dev, generator=g).half()
    w = (torch.randn(Nn, K, device=dev, generator=g) / K ** 0.5).half()
    b = torch.randn(Nn, device=dev, generator=g).half()
    r = torch.randn(M, Nn, device=dev, generator=g).half()

    def lin():
        rows, stats = [], []
        a = fused.linear(x, w, b, r, rows=rows)
        c = fused.linear(x, w, b, r, stats=stats)
        return (a, rows[0], c) + ((stats[0],) if stats else ())
    repeat("narrow-tile linear %s" % ((M, K, Nn),), lin)
# GroupNorm + SiLU inside the halo-resident convolution / inside the Winograd input transform
gn = fused.GroupNormAct(32, 128, eps=1e-6, act=True).to(dev).half().requires_grad_(False)
x = (torch.randn(2, 128, 128, 144, device=dev, generator=g) * 1.5).half().contiguous(**cl)
w_id = torch.zeros(128, 128, 3, 3, device=dev).half(); w_id[torch.arange(128), torch.arange(128), 1, 1] = 1.0
w_id = w_id.contiguous(**cl)
w = (torch.randn(128, 128, 3, 3, device=dev, generator=g) / 34.0).half().contiguous(**cl)
with torch.no_grad():
    xs = fused.conv3x3(x, w_id, gn_next=True)
stats = fused.producer_stats(xs)
repeat("GroupNorm inside the halo convolution", lambda: fused._conv_gn_in(xs, gn, None, stats, w, None, None, [])[0])
gn2 = fused.GroupNormAct(32, 1920, eps=1e-5, act=True).to(dev).half().requires_grad_(False)
a_, b_ = [torch.randn(8, c, 16, 16, device=dev, generator=g).half().contiguous(**cl) for c in (1280, 640)]
w2 = (torch.randn(1280, 1920, 3, 3, device=dev, generator=g) / 130.0).half().contiguous(**cl)
ad = torch.randn(8, 1920, device=dev, generator=g).half() * 0.3


def wino():
    xc = fused.cat_skip(a_, b_)
    return fused.conv3x3_gn(xc, gn2, ad, w2, None, None, gn_next=True)


repeat("GroupNorm inside the Winograd input transform", wino)
# in-place row softmax and its backward (the VAE's 512-channel mid attention)
sc = torch.randn(2 * 1024, 1024, device=dev, generator=g).half()
dp = torch.randn(2 * 1024, 1024, device=dev, generator=g).half()


def soft():
    pr = sc.clone()
    assert lib.gip_softmax_rows_f16(P(pr), pr.shape[0], pr.shape[1], ctypes.c_float(0.044), st()) == 0
    d = dp.clone()
    assert lib.gip_softmax_rows_backward_f16(P(pr), P(d), pr.shape[0], pr.shape[1], ctypes.c_float(1.0), st()) == 0
    return pr, d


repeat("row softmax forward + backward", soft)
# decoupled cross-attention with the second key set folded into one accumulator (D = 80 / 160: no scratch since round 5)
for (B, Hh, Nq, D) in [(4, 8, 1024, 80), (4, 8, 256, 160), (4, 8, 4096, 40)]:
    q = torch.randn(B, Nq, Hh * D, device=dev, generator=g).half()
    k, v = [torch.randn(B, 77, Hh * D, device=dev, generator=g).half() for _ in range(2)]
    k2, v2 = [torch.randn(B, 4, Hh * D, device=dev, generator=g).half() for _ in range(2)]
    repeat("two-key-set attention %s" % ((B, Hh, Nq, D),), lambda: fused.attention(q, k, v, Hh, k2, v2, 0.5))
torch.cuda.synchronize()
print("TOTAL MISMATCHES", bad)
