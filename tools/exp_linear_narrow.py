"""128 x 64 against 128 x 128 (160) tiles of the own MFMA GEMM (gip_linear_f16) on the shapes where the wide tiles leave most CUs
with at most one workgroup.  HIP-graph replay of 12 launches cycling through 12 weight copies (cold weights, as in the denoise);
outputs compared bit for bit (same K order per element)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianip_amd import _lib  # noqa: E402

lib = _lib.nn_lib()
NARROW = ctypes.c_int.in_dll(lib._lib, "gip_dbg_linear_narrow")
COPIES = 12
p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731


def graph_time(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / (reps * COPIES)


def run(M, K, N):
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(M, K, device="cuda", generator=gen).half()
    wts = [(torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5).half() for _ in range(COPIES)]
    bias = torch.randn(N, device="cuda", generator=gen).half()
    res = torch.randn(M, N, device="cuda", generator=gen).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)

    def call(w):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.gip_linear_f16(p(x), p(w), p(bias), p(res), p(out), M, K, N, 0, st) == 0
    cols, ref = [], None
    for lim in (0, 1 << 20):
        NARROW.value = lim
        call(wts[0])
        torch.cuda.synchronize()
        o = out.clone()
        ref = o if ref is None else ref
        t = min(graph_time(lambda: [call(w) for w in wts]) for _ in range(3))
        cols.append("%6.1f%s" % (t, "" if torch.equal(o, ref) else "!"))
    lib_t = min(graph_time(lambda: [torch.addmm(bias, x, w.t()) for w in wts]) for _ in range(3))
    NARROW.value = -1
    wide = 160 if N % 160 == 0 and N % 128 else 128
    tiles = ((M + 127) // 128) * ((N + wide - 1) // wide)
    return tiles, cols, lib_t


if __name__ == "__main__":
    print("M, K, N | workgroups of the wide tile | us: wide tile, 128 x 64 tile ('!' = differs), hipBLASLt addmm (no residual)")
    shapes = []
    for M in (192, 768, 3072, 12288, 49152):       # 8^2 ... 64^2 level at batch 3; 64^2 at batch 12
        for K, N in ((1280, 1280), (1280, 3840), (5120, 1280), (640, 640), (640, 1920), (2560, 640), (320, 320), (320, 960), (1280, 320), (768, 1280)):
            if M * N <= 49152 * 960:
                shapes.append((M, K, N))
    for sh in shapes:
        tiles, cols, lib_t = run(*sh)
        print("%-22s %5d   %s   %6.1f" % (sh, tiles, "  ".join(cols), lib_t), flush=True)
